repo=$(pwd); cd /tmp && export TMPDIR=/tmp
for cfg in "CREID_BENCH_CONFIG3=1" "CREID_BENCH_P=48"; do
env $cfg CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 rocprofv3 --kernel-trace -d $repo/gpurun_out/prof_tail -o t -- python $repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $repo/gpurun_out/prof_tail.log 2>&1
db=$(find $repo/gpurun_out/prof_tail -name "*.db" | head -1)
echo "== $cfg"; python $repo/tools/grid_tail.py $db train | grep -v "igemm\|wgrad_bf16" | head -14; rm -rf $repo/gpurun_out/prof_tail
done
