#!/bin/bash
# Round-5 GPU session B: the new f16-training / widening tests, the conv1 data-gradient ablation, the streamed evaluation's grid sweep
o=gpurun_out/r5c; mkdir -p $o
timeout 600 python -m pytest tests/test_f16_train_gpu.py tests/test_eval_fold_gpu.py tests/test_round2_gpu.py tests/test_ctl_step_gpu.py \
   "tests/test_backbone_gpu.py" -m gpu -q --maxfail=20 -k "f16 or float16 or deeper or lonely or base_out or conv_fwd_dgrad_wgrad or dtype2" > $o/pytest.log 2>&1
echo "pytest rc $?" >> $o/pytest.log; tail -40 $o/pytest.log
python tools/debug/dgrad_c1_probe.py 64 > $o/dgrad_c1.txt 2>&1; cat $o/dgrad_c1.txt
for w in 512 1024 2048 4096 8192; do CREID_STREAM_WGS=$w python tools/debug/stream_wgs_probe.py 2>&1 | tail -1; done | tee $o/stream_wgs.txt
