#!/bin/bash
# Round-5 GPU session A: parity suite, bench line, per-layer training table, plans "rules" vs "tuned".
#   gpurun --timeout 1200 -- bash tools/sessions/gpu_r5_session.sh
o=gpurun_out/r5b; mkdir -p $o
timeout 700 python -m pytest tests -m gpu -q --maxfail=15 --durations=10 > $o/pytest.log 2>&1
echo "pytest rc $?" >> $o/pytest.log; tail -25 $o/pytest.log
python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc $?"
bash tools/prof_train.sh r5b > $o/train_anatomy.md 2>&1
db=$(find gpurun_out/prof_r5b -name "*.db" | head -1)
python tools/train_layers.py $db > $o/train_layers.md 2>&1; tail -16 $o/train_layers.md
cp gpurun_out/prof_r5b.md $o/train_kernel_stats.md 2>/dev/null
rm -rf gpurun_out/prof_r5b
# launch plans: measured (tuned_plans.json) vs built-in rules, at the tuned batch (B = 64) and at one the file does not cover (B = 192)
for p in 16 48; do for t in 1 0; do
  CREID_BENCH_P=$p CREID_TUNED_PLANS=$t CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline \
    > $o/plans_P${p}_tuned${t}.json 2> /dev/null
  python - <<PY
import json; d=json.load(open("$o/plans_P${p}_tuned${t}.json")); print("P=$p tuned=$t", round(d["value"]), "img/s", round(d["ms_per_step"],3), "ms", d.get("plans"))
PY
done; done
