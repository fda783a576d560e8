#!/bin/bash
# Round-6 session 4: vendor yardstick alone, then the full default bench line (timed)
o=gpurun_out/r6s4; mkdir -p $o
( time python tools/vendor_step.py > $o/vendor.json 2> $o/vendor.err ) 2> $o/vendor.time; cat $o/vendor.json; tail -3 $o/vendor.time; tail -3 $o/vendor.err
( time python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/bench.time; tail -3 $o/bench.time; tail -5 $o/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6s4/bench.json"))
print(d["ms_per_step"], d["value"], d.get("launches_per_step"))
print(json.dumps(d["roofline"], indent=0)[:3000])
print(json.dumps(d.get("wgrad_family_counters")))
print(json.dumps(d["embed"].get("roofline_hbm")), json.dumps(d["embed"].get("fp32")), json.dumps(d["embed"].get("embedding_error_vs_fp32")))
print(json.dumps(d.get("extra")))
print(json.dumps(d.get("cpu_baseline")))
PY
