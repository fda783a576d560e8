"""The overlapped data-parallel step (hipGraph segments + bucketed gradient all-reduce on a side stream) against the
eager flat all-reduce: two processes share cuda:0 over gloo (the GPU box has one GPU; RCCL needs one device per rank),
identical weights on both ranks and across the three schedules after 3 steps, bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlapped_ddp_step_equals_flat_allreduce_two_ranks_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "debug", "ddp_overlap_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("DDP_OVERLAP_OK") == 2 and "DDP_OVERLAP_MISMATCH" not in out, out[-3000:]


def test_rccl_one_rank_group_matches_step_without_group():
    """First RCCL contact (backend nccl, world size 1): communicator init, flat and bucketed side-stream all-reduces between
    hipGraph segments, all_gather_into_tensor and the sharded evaluation -- all equal to the step / evaluation with no
    process group at all, bit for bit (tools/debug/rccl_world1_check.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "rccl_world1_check.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "RCCL_WORLD1_OK" in out and "RCCL_WORLD1_MISMATCH" not in out, out[-3000:]
