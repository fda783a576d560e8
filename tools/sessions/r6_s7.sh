#!/bin/bash
o=gpurun_out/r6s7; mkdir -p $o
python -m pytest tests/test_backbone_gpu.py -q -m gpu -k "operand_path or golden or noise" > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $o/pytest.log
bash tools/ab.sh "CREID_C3_AXF=" "CREID_C3_AXF=64" "CREID_C3_AXF=64,128" "CREID_C3_AXF=64,128,256" > $o/ab.txt 2>&1; cat $o/ab.txt
