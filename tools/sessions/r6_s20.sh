#!/bin/bash
# Round-6 session 20: wide-fragment LDS image of the fp32 contraction kernels (sqdist_count_f32 / sqdist_f32): parity, then A/B
o=gpurun_out/r6s20; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $o/pytest.log
for rep in 1 2; do
  echo "prev:"; CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so python tools/debug/count_probe.py 2>&1 | tail -1
  echo "new:"; python tools/debug/count_probe.py 2>&1 | tail -1
done
echo "prev:"; CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
echo "new:"; python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
