#!/bin/bash
# Round-5 GPU session L: weight-gradient plans of the other training batches re-measured with the slot-filling split counts, each A/B'd in situ
o=gpurun_out/r5q; mkdir -p $o
for b in 32 128 192 256; do
  timeout 900 python tools/tune_plans.py --batch $b --wgrad-only --merge centroids-reid_amd/tuned_plans.json --out $o/w$b.json > $o/tune_w$b.log 2>&1; tail -1 $o/tune_w$b.log
  p=$((b / 4))
  bash tools/debug/knob_ab.sh "CREID_BENCH_P=$p" "CREID_BENCH_P=$p CREID_TUNED_PLANS=$o/w$b.json" "CREID_BENCH_P=$p" "CREID_BENCH_P=$p CREID_TUNED_PLANS=$o/w$b.json"
done
