#!/bin/bash
# Round-5 GPU session G: eval-mode plans for the embedding batches 192 / 384 / 512 (additive), A/B
o=gpurun_out/r5j; mkdir -p $o
cur=centroids-reid_amd/tuned_plans.json
for b in 192 384 512; do
  timeout 600 python tools/tune_plans.py --batch $b --fwd-only --merge $cur --out $o/e$b.json > $o/tune_e$b.log 2>&1; tail -1 $o/tune_e$b.log
  timeout 600 python tools/tune_plans.py --batch $b --fwd-only --pp-only --merge $o/e$b.json --out $o/ep$b.json > $o/tune_ep$b.log 2>&1; tail -1 $o/tune_ep$b.log
  python tools/merge_plans.py $cur $o/ep$b.json $o/m$b.json
  cur=$o/m$b.json
done
cp $cur $o/plans_final.json
for t in centroids-reid_amd/tuned_plans.json $o/plans_final.json; do echo "embed plans=$t"; CREID_TUNED_PLANS=$t python tools/debug/embed_batch_sweep.py 2>&1 | grep "img/s"; done | tee $o/ab_embed.txt
for p in 16 48; do CREID_BENCH_P=$p CREID_TUNED_PLANS=$o/plans_final.json CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train P=$p', round(d['value']), round(d['ms_per_step'],3))"; done
