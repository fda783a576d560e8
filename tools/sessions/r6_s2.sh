#!/bin/bash
# Round-6 session 2: fused heads -- parity tests, then the kernel trace of the step
o=gpurun_out/r6s2; mkdir -p $o
python -m pytest tests/test_heads_fused_gpu.py tests/test_ctl_step_gpu.py tests/test_bench_path_gpu.py tests/test_f16_train_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $o/pytest.log
bash tools/prof_train.sh r6s2 > $o/train_step_anatomy.md 2>&1; head -20 $o/train_step_anatomy.md
db=$(find gpurun_out/prof_r6s2 -name "*.db" | head -1)
python tools/step_sequence.py $db > $o/step_sequence.txt 2>&1; grep -n "heads_stage\|gap_\|adam\|sgd\|Fill\|copyBuffer\|weight_prep\|elementwise" $o/step_sequence.txt
rm -rf gpurun_out/prof_r6s2
