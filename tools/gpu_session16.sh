#!/bin/bash
mkdir -p gpurun_out
cp centroids-reid_amd/tuned_plans.json gpurun_out/tuned_plans_b64.json
python tools/tune_plans.py --batch 64 --h 256 --w 128 --out gpurun_out/tuned_plans_b64.json --merge gpurun_out/tuned_plans_b64.json > gpurun_out/s16_tune_b64.log 2>&1
tail -70 gpurun_out/s16_tune_b64.log
