#!/bin/bash
# Round-5 GPU session D: launch plans for more batches of the 256 x 128 workloads (training B = 32 / 128 / 256, embedding B = 64 / 256),
# each tuned on top of the previous file and A/B'd in situ against the shipped file (which has no entries for these batches = rules).
o=gpurun_out/r5g; mkdir -p $o
cur=centroids-reid_amd/tuned_plans.json
for b in 32 128 256; do
  timeout 600 python tools/tune_plans.py --batch $b --merge $cur --out $o/plans_t$b.json > $o/tune_t$b.log 2>&1; tail -1 $o/tune_t$b.log
  cur=$o/plans_t$b.json
done
for b in 64 256; do
  timeout 600 python tools/tune_plans.py --batch $b --fwd-only --merge $cur --out $o/plans_e$b.json > $o/tune_e$b.log 2>&1; tail -1 $o/tune_e$b.log
  cur=$o/plans_e$b.json
  timeout 600 python tools/tune_plans.py --batch $b --fwd-only --pp-only --merge $cur --out $o/plans_ep$b.json > $o/tune_ep$b.log 2>&1; tail -1 $o/tune_ep$b.log
  cur=$o/plans_ep$b.json
done
cp $cur $o/plans_all.json
for p in 8 32 64 16; do for t in centroids-reid_amd/tuned_plans.json $o/plans_all.json; do
  CREID_BENCH_P=$p CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/tmp.json 2>/dev/null
  python -c "import json; d=json.load(open('$o/tmp.json')); print('train B=%d' % ($p*4), '$t'.split('/')[-1], round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done | tee $o/ab_train.txt
for t in centroids-reid_amd/tuned_plans.json $o/plans_all.json; do echo "embed plans=$t"; CREID_TUNED_PLANS=$t python tools/debug/embed_batch_sweep.py 2>&1 | grep "img/s"; done | tee $o/ab_embed.txt
