#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_pl_branch_gpu.py tests/test_round2_gpu.py tests/test_stream_eval_gpu.py -m gpu -q > gpurun_out/s6_pytest.log 2>&1; tail -6 gpurun_out/s6_pytest.log
python tools/debug/pl_stub_check.py 2>&1 | grep -v amdgpu | tail -8
CREID_DIST_BACKEND=gloo CREID_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/s6_ddp2.json 2> gpurun_out/s6_ddp2.err
echo "ddp2 rc $?"; tail -3 gpurun_out/s6_ddp2.err | cut -c1-300; cut -c1-700 gpurun_out/s6_ddp2.json
CREID_FORCE_DIST=1 CREID_BENCH_NO_EVAL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s6_force_dist.json 2> gpurun_out/s6_force_dist.err
echo "force-dist rc $?"; cut -c1-400 gpurun_out/s6_force_dist.json
