#!/bin/bash
# One GPU session that regenerates everything under profiles/: parity suite, smoke, the bench line, step + embedding-forward
# anatomy, the evaluation kernels' trace, counter passes.   gpurun --timeout 2400 -- bash tools/gpu_full_session.sh
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > gpurun_out/full_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/full_pytest.log
tail -14 gpurun_out/full_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.log 2>&1; tail -2 gpurun_out/full_smoke.log
python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
echo "bench rc $?"
bash tools/prof_train.sh full > gpurun_out/full_anatomy.md 2>&1
head -24 gpurun_out/full_anatomy.md
cp gpurun_out/prof_full.md gpurun_out/full_train_kernel_stats.md 2>/dev/null
bash tools/prof_embed.sh fulle > gpurun_out/full_embed_anatomy.md 2>&1
tail -75 gpurun_out/full_embed_anatomy.md
# the evaluation half of the metric: kernel trace of the same bench command (eval.roofline.frac is reproducible from it)
repo=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_fullev -o ev -- \
   python $repo/bench.py --workload eval --steps 5 --warmup 2 --no-cpu-baseline > $repo/gpurun_out/prof_fullev.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_fullev -name "*.db" | head -1) gpurun_out/full_eval_kernel_stats.md > /dev/null
head -16 gpurun_out/full_eval_kernel_stats.md
rm -rf gpurun_out/prof_full gpurun_out/prof_fulle gpurun_out/prof_fullev
bash tools/pmc_run.sh > gpurun_out/full_pmc.log 2>&1; tail -8 gpurun_out/full_pmc.log
