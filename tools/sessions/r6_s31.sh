#!/bin/bash
# Round-6 session 31: label upload through a pinned staging buffer + one workspace allocation for the device index: tests, end-to-end A/B
o=gpurun_out/r6s31; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py tests/test_round2_gpu.py -q -m gpu -x > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $o/pytest.log
for rep in 1 2 3; do
    python bench.py --workload eval --steps 20 --warmup 5 --no-cpu-baseline > $o/eval_$rep.json 2> $o/eval.err
    python - <<PY
import json
d = json.load(open("$o/eval_$rep.json"))
print(round(d["value"] / 1e9, 2), "G pairs/s", round(d["ms_per_step"], 4), "ms", d["stages_ms"], round(d["roofline"]["frac"], 3), "first", round(d["first_call_ms"], 3))
PY
done
