#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > gpurun_out/s4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/s4_pytest.log
tail -14 gpurun_out/s4_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s4_smoke.log 2>&1; tail -2 gpurun_out/s4_smoke.log
python bench.py > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err
echo "bench rc $?"
bash tools/prof_train.sh s4 > gpurun_out/s4_anatomy.md 2>&1
head -24 gpurun_out/s4_anatomy.md
bash tools/prof_embed.sh s4e > gpurun_out/s4_embed_anatomy.md 2>&1
tail -75 gpurun_out/s4_embed_anatomy.md
rm -rf gpurun_out/prof_s4 gpurun_out/prof_s4e
bash tools/pmc_run.sh > gpurun_out/s4_pmc.log 2>&1; tail -8 gpurun_out/s4_pmc.log
