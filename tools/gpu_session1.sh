#!/bin/bash
# round-3 GPU session 1: parity suite, full bench line, step anatomy
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=25 --durations=15 > gpurun_out/s1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/s1_pytest.log
tail -30 gpurun_out/s1_pytest.log
python bench.py > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
echo "bench rc $?"
tail -5 gpurun_out/s1_bench.err
cat gpurun_out/s1_bench.json | head -c 6000
bash tools/prof_train.sh s1 > gpurun_out/s1_anatomy.md 2>&1
head -40 gpurun_out/s1_anatomy.md
