#!/bin/bash
# rocprofv3 kernel trace of the training bench (run on the GPU box):  bash tools/prof_train.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>/ and prints the per-kernel table normalised per step.
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
env "$@" CREID_BENCH_NO_EVAL=1 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_$tag -o train -- \
  python $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $repo/gpurun_out/prof_$tag.log 2>&1
cd $repo
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/prof_$tag.md 60 > /dev/null
python tools/step_anatomy.py $db
