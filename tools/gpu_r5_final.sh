#!/bin/bash
# Round-5 final GPU session: everything under profiles/r05_* from one box.   gpurun --timeout 2400 -- bash tools/gpu_r5_final.sh
export CREID_ROUND=r05
o=gpurun_out/r5f; mkdir -p $o
python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > $o/pytest.log 2>&1
echo "pytest rc $?" >> $o/pytest.log; tail -14 $o/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -2 $o/smoke.log
python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc $?"
bash tools/prof_train.sh r5f > $o/train_step_anatomy.md 2>&1; head -22 $o/train_step_anatomy.md
db=$(find gpurun_out/prof_r5f -name "*.db" | head -1)
python tools/train_layers.py $db > $o/train_layers.md 2>&1; tail -14 $o/train_layers.md
cp gpurun_out/prof_r5f.md $o/train_kernel_stats.md 2>/dev/null
bash tools/prof_embed.sh r5fe > $o/embed_anatomy.md 2>&1; tail -4 $o/embed_anatomy.md
repo=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_r5fev -o ev -- \
   python $repo/bench.py --workload eval --steps 5 --warmup 2 --no-cpu-baseline > $repo/gpurun_out/prof_r5fev.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r5fev -name "*.db" | head -1) $o/eval_kernel_stats.md > /dev/null; head -14 $o/eval_kernel_stats.md
rm -rf gpurun_out/prof_r5f gpurun_out/prof_r5fe gpurun_out/prof_r5fev
# HBM-side traffic of the streamed evaluation's contraction on the configs[3] shard, old grid rule vs the shipped one
for t in 1000 8; do
  (cd /tmp && export TMPDIR=/tmp && CREID_STREAM_TPER=$t rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $repo/gpurun_out/pmc_stream_$t -o p -- \
     python $repo/tools/pmc_stream.py > $repo/gpurun_out/pmc_stream_$t.log 2>&1)
  python tools/pmc_summary.py $(find gpurun_out/pmc_stream_$t -name "*.db" | head -1) $o/pmc_stream_tper$t.json 2>&1 | grep sqdist_count
  rm -rf gpurun_out/pmc_stream_$t
done
bash tools/pmc_run.sh > $o/pmc.log 2>&1; tail -6 $o/pmc.log
python tools/pmc_report.py > $o/pmc_report.log 2>&1; tail -5 $o/pmc_report.log
cp profiles/r05_pmc_traffic.json profiles/r05_pmc_summary.md $o/ 2>/dev/null
