#!/bin/bash
# Round-6 session 27: does the count kernel need the gallery tiles shared inside an XCD?  query-tile-major order (CREID_STREAM_ORDER=1: the
# 14 workgroups of a query tile side by side, every workgroup streaming its own gallery range) against the shipped split-major order
for o in 0 1; do
  CREID_STREAM_ORDER=$o python tools/debug/count_probe.py 2>&1 | tail -1
  CREID_STREAM_ORDER=$o python tools/debug/count_probe.py 3000 15000 2>&1 | tail -1
done
CREID_STREAM_ORDER=1 python tools/debug/stream_wgs_probe.py 2>&1 | tail -1
