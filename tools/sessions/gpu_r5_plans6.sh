#!/bin/bash
# Round-5 GPU session I: the all-waves-multiply kernel for the TRAINING forward of the new batches (128 / 192 / 256), additive, A/B
o=gpurun_out/r5m; mkdir -p $o
cur=centroids-reid_amd/tuned_plans.json
for b in 128 192 256; do
  timeout 900 python tools/tune_plans.py --batch $b --pp-only --merge $cur --out $o/pp$b.json > $o/tune_pp$b.log 2>&1; tail -1 $o/tune_pp$b.log
  python tools/merge_plans.py $cur $o/pp$b.json $o/m$b.json
  cur=$o/m$b.json
done
cp $cur $o/plans_final.json
for p in 32 48 64 16; do for t in centroids-reid_amd/tuned_plans.json $o/plans_final.json; do
  CREID_BENCH_P=$p CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/tmp.json 2>$o/tmp.err || tail -3 $o/tmp.err
  python -c "import json; d=json.load(open('$o/tmp.json')); print('train B=%d' % ($p*4), '$t'.split('/')[-1], round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done | tee $o/ab_train.txt
