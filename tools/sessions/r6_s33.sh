#!/bin/bash
o=gpurun_out/r6s33; mkdir -p $o
( time python bench.py --workload eval --steps 10 --warmup 3 --no-cpu-baseline > $o/eval.json 2> $o/eval.err ) 2> $o/time.txt; tail -3 $o/time.txt; tail -3 $o/eval.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6s33/eval.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"])); print(json.dumps(d["materialised"]["roofline"]))
PY
