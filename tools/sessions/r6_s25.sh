#!/bin/bash
# Round-6 session 25: one global load per MFMA gap; parity; A/B against the first commit of the day (c1) on one box
o=gpurun_out/r6s25; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $o/pytest.log
for rep in 1 2; do
  echo "c1:"; CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_c1.so python tools/debug/count_probe.py 2>&1 | tail -1
  echo "new:"; python tools/debug/count_probe.py 2>&1 | tail -1
done
echo "c1:"; CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_c1.so python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
echo "new:"; python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
export CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_abl.so
for abl in 0 64 2; do
  e=$((1 + abl))
  echo "ABL=$abl (NOEPI=$e)"
  CREID_STREAM_NOEPI=$e python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1 | cut -c1-90
done
