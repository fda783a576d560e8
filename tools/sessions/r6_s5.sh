#!/bin/bash
o=gpurun_out/r6s5; mkdir -p $o
python -m pytest tests/test_heads_fused_gpu.py tests/test_bench_path_gpu.py tests/test_f16_train_gpu.py tests/test_ctl_step_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $o/pytest.log
bash tools/ab.sh "CREID_HEADS_BNRED=0" "CREID_HEADS_BNRED=1" > $o/ab.txt 2>&1; cat $o/ab.txt
