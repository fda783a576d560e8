#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ddp_overlap_gpu.py -m gpu -q > gpurun_out/s7_pytest.log 2>&1; tail -15 gpurun_out/s7_pytest.log
CREID_FORCE_DIST=1 CREID_BENCH_NO_INSITU=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s7_force_dist.json 2> gpurun_out/s7_force_dist.err
echo "force-dist rc $?"; cut -c1-300 gpurun_out/s7_force_dist.json; grep -v amdgpu gpurun_out/s7_force_dist.err | tail -5 | cut -c1-300
