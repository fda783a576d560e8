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
