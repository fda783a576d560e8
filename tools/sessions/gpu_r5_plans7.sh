#!/bin/bash
# Round-5 GPU session J: re-measure the B = 64 training plans against the current kernels (replacing merge), A/B in situ twice
o=gpurun_out/r5n; mkdir -p $o
timeout 900 python tools/tune_plans.py --batch 64 --merge centroids-reid_amd/tuned_plans.json --out $o/t64.json > $o/tune_t64.log 2>&1; tail -1 $o/tune_t64.log
timeout 900 python tools/tune_plans.py --batch 64 --pp-only --merge $o/t64.json --out $o/t64pp.json > $o/tune_t64pp.log 2>&1; tail -1 $o/tune_t64pp.log
for r in 1 2; do for t in centroids-reid_amd/tuned_plans.json $o/t64.json $o/t64pp.json; do
  CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $o/tmp.json 2>$o/tmp.err || tail -3 $o/tmp.err
  python -c "import json; d=json.load(open('$o/tmp.json')); print('train B=64', '$t'.split('/')[-1], round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"
done; done | tee $o/ab_train.txt
