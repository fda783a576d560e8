#!/bin/bash
# Round-5 GPU session K: weight-gradient plans of B = 64 re-measured with the slot-filling split counts, A/B in situ (twice)
o=gpurun_out/r5p; mkdir -p $o
timeout 900 python tools/tune_plans.py --batch 64 --wgrad-only --merge centroids-reid_amd/tuned_plans.json --out $o/w64.json > $o/tune_w64.log 2>&1; tail -3 $o/tune_w64.log
bash tools/debug/knob_ab.sh A=0 CREID_TUNED_PLANS=$o/w64.json A=1 CREID_TUNED_PLANS=$o/w64.json
