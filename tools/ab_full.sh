#!/bin/bash
# A/B of env settings on the whole default bench line (run on the GPU box): bash tools/ab_full.sh "A=1" "A=0" ...
# prints the training step, the embedding forward (R50 bs128 and IBN-a 320x320 bs256) and the end-to-end evaluation per setting
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg CREID_BENCH_NO_INSITU=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('embed',{}); ev=d.get('eval',{})
print('$cfg : train %.3f ms | embed %.0f img/s | ibn %.0f img/s | eval %.3f ms' % (d['ms_per_step'], e.get('value',0), e.get('configs3_embedding_half',{}).get('value',0), ev.get('ms_per_step',0)))"
  done
done
