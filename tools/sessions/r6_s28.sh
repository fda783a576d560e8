#!/bin/bash
# Round-6 session 28: equal-run work split (mode 1) of the count kernel: parity in both modes, then timing against mode 0
o=gpurun_out/r6s28; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu -x > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $o/pytest.log
for b in 0 1; do
  CREID_STREAM_BALANCE=$b python tools/debug/count_probe.py 2>&1 | tail -1
  CREID_STREAM_BALANCE=$b python tools/debug/count_probe.py 3000 15000 2>&1 | tail -1
  CREID_STREAM_BALANCE=$b python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1
done
python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
