#!/bin/bash
# Round-5 GPU session E: re-tune the B = 128 eval-mode plans on top of the candidate file, A/B everything in situ
o=gpurun_out/r5h; mkdir -p $o
cur=centroids-reid_amd/plans_candidate.json
timeout 600 python tools/tune_plans.py --batch 128 --fwd-only --merge $cur --out $o/p1.json > $o/tune_e128.log 2>&1; tail -1 $o/tune_e128.log
timeout 600 python tools/tune_plans.py --batch 128 --fwd-only --pp-only --merge $o/p1.json --out $o/p2.json > $o/tune_ep128.log 2>&1; tail -1 $o/tune_ep128.log
for t in centroids-reid_amd/tuned_plans.json centroids-reid_amd/plans_candidate.json $o/p2.json; do echo "embed plans=$t"; CREID_TUNED_PLANS=$t python tools/debug/embed_batch_sweep.py 2>&1 | grep "img/s"; done | tee $o/ab_embed.txt
for p in 8 16; do for t in centroids-reid_amd/tuned_plans.json $o/p2.json; do
  CREID_BENCH_P=$p CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/tmp.json 2>$o/tmp.err || tail -3 $o/tmp.err
  python -c "import json; d=json.load(open('$o/tmp.json')); print('train B=%d' % ($p*4), '$t'.split('/')[-1], round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done | tee $o/ab_train.txt
