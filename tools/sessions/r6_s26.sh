#!/bin/bash
# Round-6 session 26: parity after the address fix; is the 8 KB row pitch (D = 2048) camping on cache channels?  D = 2048 / 2064 / 2112 / 2176
o=gpurun_out/r6s26; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $o/pytest.log
for D in 2048 2064 2112 2176 1984; do
  CREID_STREAM_NOEPI=1 python tools/debug/count_probe.py 2048 20480 $D 2>&1 | tail -1
done
