#!/bin/bash
# Round-6 session 29: default mode selection; eval tests; end-to-end eval line of the bench
o=gpurun_out/r6s29; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu -x > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $o/pytest.log
python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
python bench.py --workload eval --steps 10 --warmup 3 --no-cpu-baseline > $o/eval.json 2> $o/eval.err; tail -2 $o/eval.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6s29/eval.json"))
print(json.dumps(d)[:3000])
PY
