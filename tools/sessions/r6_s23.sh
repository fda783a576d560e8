#!/bin/bash
# Round-6 session 23: four-phase k-loop with two staged k-tiles: parity, A/B against the previous build, ablations
o=gpurun_out/r6s23; mkdir -p $o
python -m pytest tests/test_eval_gpu.py tests/test_stream_eval_gpu.py tests/test_centroid_eval_gpu.py tests/test_parity_full_size_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $o/pytest.log
for rep in 1 2; do
  echo "prev:"; CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so python tools/debug/count_probe.py 2>&1 | tail -1
  echo "new:"; python tools/debug/count_probe.py 2>&1 | tail -1
done
echo "new:"; python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
export CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_abl.so
for abl in 0 2 4 8 14 16 30; do
  e=$((1 + abl))
  echo "ABL=$abl (NOEPI=$e)"
  CREID_STREAM_NOEPI=$e python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1 | cut -c1-90
done
