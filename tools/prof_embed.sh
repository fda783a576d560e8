#!/bin/bash
# rocprofv3 kernel trace of `bench.py --inner-trace` (training step replays, then embedding-forward replays) on the GPU box:
#   bash tools/prof_embed.sh <tag>   -> gpurun_out/prof_<tag>/ , prints the per-launch table of one embedding forward
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
env "$@" CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_$tag -o inner -- \
  python $repo/bench.py --inner-trace > $repo/gpurun_out/prof_$tag.log 2>&1
cd $repo
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/embed_anatomy.py $db 128 256 128
