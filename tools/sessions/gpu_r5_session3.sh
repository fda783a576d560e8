#!/bin/bash
# Round-5 GPU session C: re-run of the fixed tests, gallery tiles per workgroup of the streamed evaluation, launch plans for an untuned batch (B = 192)
o=gpurun_out/r5d; mkdir -p $o
timeout 600 python -m pytest tests/test_f16_train_gpu.py tests/test_eval_fold_gpu.py tests/test_backbone_gpu.py tests/test_stream_eval_gpu.py -m gpu -q --maxfail=20 \
   -k "f16 or deeper or stream" > $o/pytest.log 2>&1
echo "pytest rc $?" >> $o/pytest.log; tail -12 $o/pytest.log
for t in 1 2 4 8 16 32; do CREID_STREAM_TPER=$t python tools/debug/stream_wgs_probe.py 2>&1 | tail -1; done | tee $o/stream_tper.txt
timeout 900 python tools/tune_plans.py --batch 192 --merge centroids-reid_amd/tuned_plans.json --out $o/plans_b192.json > $o/tune_b192.log 2>&1; tail -5 $o/tune_b192.log
for t in 0 $o/plans_b192.json; do
  CREID_BENCH_P=48 CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/p48.json 2>/dev/null
  python -c "import json; d=json.load(open('$o/p48.json')); print('P=48 plans=$t', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms', d.get('plans'))"
done | tee $o/plans_b192.txt
