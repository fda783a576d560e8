repo=$(pwd); cd /tmp && export TMPDIR=/tmp
CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 rocprofv3 --kernel-trace -d $repo/gpurun_out/prof_tail -o inner -- python $repo/bench.py --inner-trace > $repo/gpurun_out/prof_tail.log 2>&1
cd $repo; db=$(find gpurun_out/prof_tail -name "*.db" | head -1)
python tools/grid_tail.py $db train; python tools/grid_tail.py $db embed; rm -rf gpurun_out/prof_tail
