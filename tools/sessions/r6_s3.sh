#!/bin/bash
# Round-6 session 3: tail items (weight copies as a graph branch, Adam self-advancing, forward counter, fills) -- tests + A/B + trace
o=gpurun_out/r6s3; mkdir -p $o
python -m pytest tests/test_heads_fused_gpu.py tests/test_heads_gpu.py tests/test_ctl_step_gpu.py tests/test_bench_path_gpu.py tests/test_f16_train_gpu.py tests/test_round2_gpu.py "tests/test_conv_pipe_gpu.py::test_knobs_are_read_once_in_production_mode" tests/test_ddp_overlap_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $o/pytest.log
bash tools/ab.sh "CREID_WPREP_SIDE=0" "CREID_WPREP_SIDE=1" > $o/ab.txt 2>&1; cat $o/ab.txt
bash tools/prof_train.sh r6s3 > $o/train_step_anatomy.md 2>&1; head -18 $o/train_step_anatomy.md
db=$(find gpurun_out/prof_r6s3 -name "*.db" | head -1)
python tools/step_sequence.py $db > $o/step_sequence.txt 2>&1; grep -n "heads_stage\|gap_\|adam\|sgd\|Fill\|copyBuffer\|weight_prep\|elementwise\|image_pad" $o/step_sequence.txt; head -8 $o/step_sequence.txt
rm -rf gpurun_out/prof_r6s3
