#!/bin/bash
o=gpurun_out/r6s36; mkdir -p $o
( time python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/time.txt; echo "rc $?"; tail -3 $o/time.txt; tail -2 $o/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6s36/bench.json"))
print(d["ms_per_step"], d["value"]); print(json.dumps(d["extra"]["vendor_yardstick"].get("eval"))); print(d["eval"]["value"], d["eval"]["roofline"]["traffic_source"][:40], d["roofline"]["traffic_source"][:40])
PY
python -m pytest tests/test_bench_path_gpu.py tests/test_ddp_overlap_gpu.py -q -m gpu -x 2>&1 | tail -2
