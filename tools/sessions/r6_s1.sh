#!/bin/bash
# Round-6 session 1: baseline of the box (train bench, kernel trace, ordered sequence of one step)
o=gpurun_out/r6s1; mkdir -p $o
CREID_BENCH_NO_EVAL=1 python bench.py --no-cpu-baseline > $o/bench_train.json 2> $o/bench_train.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$o/bench_train.json'));print(d['ms_per_step'],d['value'])"
bash tools/prof_train.sh r6s1 > $o/train_step_anatomy.md 2>&1; head -20 $o/train_step_anatomy.md
db=$(find gpurun_out/prof_r6s1 -name "*.db" | head -1)
python tools/step_sequence.py $db > $o/step_sequence.txt 2>&1; tail -1 $o/step_sequence.txt
rm -rf gpurun_out/prof_r6s1
