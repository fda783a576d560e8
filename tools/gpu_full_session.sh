#!/bin/bash
# One GPU session that regenerates everything under profiles/: parity suite, smoke, the bench line, step + embedding-forward
# anatomy, counter passes.   gpurun --timeout 2400 -- bash tools/gpu_full_session.sh
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > gpurun_out/full_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/full_pytest.log
tail -14 gpurun_out/full_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.log 2>&1; tail -2 gpurun_out/full_smoke.log
python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
echo "bench rc $?"
bash tools/prof_train.sh full > gpurun_out/full_anatomy.md 2>&1
head -24 gpurun_out/full_anatomy.md
bash tools/prof_embed.sh fulle > gpurun_out/full_embed_anatomy.md 2>&1
tail -75 gpurun_out/full_embed_anatomy.md
rm -rf gpurun_out/prof_full gpurun_out/prof_fulle
bash tools/pmc_run.sh > gpurun_out/full_pmc.log 2>&1; tail -8 gpurun_out/full_pmc.log
