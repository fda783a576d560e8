#!/bin/bash
export CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_abl.so
for abl in 0 64 2; do
  e=$((1 + abl))
  echo "ABL=$abl (NOEPI=$e)"
  CREID_STREAM_NOEPI=$e python tools/debug/count_probe.py 2048 20480 2>&1 | tail -1 | cut -c1-90
done
