#!/bin/bash
# Round-6 session 22: k-loop ablations of the count kernel (ablation build), exact-fit grid 2048 x 20480; CREID_STREAM_NOEPI = 1 (no epilogue) + 2 x ABL bits
export CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_abl.so
for abl in 0 2 4 6 8 14 16 30 32; do
  e=$((1 + 2 * abl / 2 * 2 / 2 * 1)); e=$((1 + abl))
  echo "ABL=$abl (NOEPI=$e)"
  CREID_STREAM_NOEPI=$e python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1 | cut -c1-90
  CREID_STREAM_NOEPI=$e CREID_STREAM_WGS=256 python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1 | cut -c1-90
done
