#!/bin/bash
# Round-6 session 30: end-to-end evaluation, round-5 library against the new one on one box (two interleaved repeats)
o=gpurun_out/r6s30; mkdir -p $o
for rep in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so; else unset CREID_LIB_PATH; fi
    python bench.py --workload eval --steps 20 --warmup 5 --no-cpu-baseline > $o/eval_${lib}_$rep.json 2> $o/eval.err
    python - <<PY
import json
d = json.load(open("$o/eval_${lib}_$rep.json"))
print("$lib", round(d["value"] / 1e9, 2), "G pairs/s", round(d["ms_per_step"], 4), "ms", d["stages_ms"], round(d["roofline"]["frac"], 3), "materialised", round(d["materialised"]["ms_per_step"], 4))
PY
  done
done
