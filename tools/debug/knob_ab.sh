#!/bin/bash
# same-box A/B of environment knobs on the captured B = 64 training step:  bash tools/debug/knob_ab.sh "K=V [K=V ...]" ...
run() { env $1 CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
for cfg in "$@"; do run "$cfg"; done
