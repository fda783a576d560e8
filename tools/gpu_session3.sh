#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_backbone_gpu.py::test_whole_network_gradient_error_is_at_the_fp32_noise_floor tests/test_backbone_gpu.py::test_dual_apply_equals_separate_downsample_bn_fp32 tests/test_round2_gpu.py -m gpu -q -s > gpurun_out/s3_pytest.log 2>&1; grep -n "conv-weight\|worst tensor\|passed\|failed" gpurun_out/s3_pytest.log
python tools/debug/bm256_probe.py 128 > gpurun_out/s3_bm256_b128.txt 2>&1; grep -c "eq=True" gpurun_out/s3_bm256_b128.txt; grep "eq=False" gpurun_out/s3_bm256_b128.txt | head -3
cp centroids-reid_amd/tuned_plans.json gpurun_out/tuned_plans_new.json
python tools/tune_plans.py --fwd-only --batch 128 --h 256 --w 128 --out gpurun_out/tuned_plans_new.json --merge gpurun_out/tuned_plans_new.json > gpurun_out/s3_tune_b128.log 2>&1; tail -25 gpurun_out/s3_tune_b128.log
python tools/tune_plans.py --fwd-only --batch 256 --h 320 --w 320 --out gpurun_out/tuned_plans_new.json --merge gpurun_out/tuned_plans_new.json > gpurun_out/s3_tune_ibn.log 2>&1; tail -25 gpurun_out/s3_tune_ibn.log
