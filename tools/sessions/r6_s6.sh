#!/bin/bash
o=gpurun_out/r6s6; mkdir -p $o
python -m pytest tests/test_backbone_gpu.py tests/test_heads_fused_gpu.py tests/test_bench_path_gpu.py tests/test_round2_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $o/pytest.log
bash tools/ab.sh "CREID_DS_REDUCE2=0" "CREID_DS_REDUCE2=1" > $o/ab.txt 2>&1; cat $o/ab.txt
