#!/bin/bash
# Round-5 GPU session F: in-situ A/B of the shipped (additively merged) plan file against the built-in rules
o=gpurun_out/r5i; mkdir -p $o
for p in 8 16 32 48 64; do for t in 0 1; do
  CREID_BENCH_P=$p CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/tmp.json 2>$o/tmp.err || tail -3 $o/tmp.err
  python -c "import json; d=json.load(open('$o/tmp.json')); print('train B=%d' % ($p*4), 'plans' if $t else 'rules', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done | tee $o/ab_train.txt
for t in 0 1; do echo "embed tuned=$t"; CREID_TUNED_PLANS=$t python tools/debug/embed_batch_sweep.py 2>&1 | grep "img/s"; done | tee $o/ab_embed.txt
