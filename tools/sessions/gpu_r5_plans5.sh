#!/bin/bash
# Round-5 GPU session H: eval-mode plans for the 320 x 320 embedding batches 128 / 512 (configs[3] uses 256, tuned in round 3/4), A/B
o=gpurun_out/r5l; mkdir -p $o
cur=centroids-reid_amd/tuned_plans.json
for b in 128 512; do
  timeout 900 python tools/tune_plans.py --batch $b --h 320 --w 320 --fwd-only --merge $cur --out $o/e$b.json > $o/tune_e$b.log 2>&1; tail -1 $o/tune_e$b.log
  timeout 900 python tools/tune_plans.py --batch $b --h 320 --w 320 --fwd-only --pp-only --merge $o/e$b.json --out $o/ep$b.json > $o/tune_ep$b.log 2>&1; tail -1 $o/tune_ep$b.log
  python tools/merge_plans.py $cur $o/ep$b.json $o/m$b.json
  cur=$o/m$b.json
done
cp $cur $o/plans_final.json
for t in centroids-reid_amd/tuned_plans.json $o/plans_final.json; do echo "plans=$t"; CREID_TUNED_PLANS=$t python tools/debug/embed_batch_sweep.py resnet50_ibn_a 320 320 128 256 512 2>&1 | grep "img/s"; done | tee $o/ab_embed320.txt
