#!/bin/bash
# A/B of env settings on the training bench (run on the GPU box): bash tools/ab.sh "A=1 B=2" "A=0" ...   (each config twice, interleaved)
for rep in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
    echo "$cfg : $ms ms"
  done
done
