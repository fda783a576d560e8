#!/bin/bash
mkdir -p gpurun_out
python tools/debug/bm256_probe.py 64 > gpurun_out/s2_bm256_b64.txt 2>&1; tail -25 gpurun_out/s2_bm256_b64.txt
python tools/debug/bm256_probe.py 128 > gpurun_out/s2_bm256_b128.txt 2>&1; tail -25 gpurun_out/s2_bm256_b128.txt
python tools/debug/bm256_probe.py 256 320 320 > gpurun_out/s2_bm256_ibn.txt 2>&1; tail -25 gpurun_out/s2_bm256_ibn.txt
python -m pytest tests/test_backbone_gpu.py tests/test_round2_gpu.py tests/test_eval_fold_gpu.py -m gpu -q > gpurun_out/s2_pytest.log 2>&1; tail -5 gpurun_out/s2_pytest.log
for v in 0 1 2 3; do CREID_BN_APPLY_DRY=$v CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BN_APPLY_DRY=$v ms_per_step', d['ms_per_step'])"; done
for v in 1 0; do CREID_DUAL_APPLY=$v CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DUAL_APPLY=$v ms_per_step', d['ms_per_step'])"; done
