#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_round2_gpu.py -m gpu -q > gpurun_out/s11_pytest.log 2>&1; tail -8 gpurun_out/s11_pytest.log
python bench.py --workload eval > gpurun_out/s11_eval.json 2> gpurun_out/s11_eval.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/s11_eval.json'))
print('e2e', d['value'], d['ms_per_step']); m=d['materialised']; print('mat', m['value'], m['ms_per_step'], m['stages_ms'])"
