"""The overlapped data-parallel step (hipGraph segments + bucketed gradient all-reduce on a side stream) against the
eager flat all-reduce: two processes share cuda:0 over gloo (the GPU box has one GPU; RCCL needs one device per rank),
identical weights on both ranks and across the three schedules after 3 steps, bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
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


def test_bench_data_parallel_path_over_one_rank_rccl_group_matches_single_process_bench():
    """`bench.py` itself on the data-parallel path (hipGraph segments + bucketed RCCL all-reduces on a side stream) with a
    one-rank `nccl` group at the REAL benchmark size, against the plain single-process bench: same loss after the same steps.
    (This run is what found that hipGraph capture in the default global mode aborts the process as soon as the RCCL watchdog
    thread polls an event -- every capture of the bench now uses the thread-local mode.)"""
    import json
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CREID_BENCH_NO_EVAL="1", CREID_BENCH_NO_INSITU="1", CREID_DETERMINISTIC="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        base.pop(k, None)
    losses = {}
    for mode, extra in (("single", {}), ("rccl1", {"CREID_FORCE_DIST": "1", "MASTER_PORT": "29541"})):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                           cwd=ROOT, env=dict(base, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (mode, (r.stdout + r.stderr)[-3000:])
        out_lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
        assert len(out_lines) == 1, out_lines                      # the bench contract: ONE line on stdout (RCCL's banner included)
        line = json.loads(out_lines[0])
        assert line["hip_graph"] and line["n_gpus"] == 1
        losses[mode] = (line["final_loss"], line["ms_per_step"])
    assert losses["single"][0] == losses["rccl1"][0], losses
    # three graph replays + four host-issued collectives per step instead of one replay: a bounded overhead
    assert losses["rccl1"][1] < 1.5 * losses["single"][1] + 1.0, losses


def test_bench_gpus_2_contract_two_ranks_one_gpu():
    """The exact command the scaling driver issues -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`
    -- on a one-GPU box (both ranks on cuda:0, gloo for the collectives: CREID_DIST_BACKEND=gloo CREID_SINGLE_DEVICE=1): rc 0,
    exactly ONE stdout line, the N = 2 control flow of every object in it (global batch, dp2, whole-job embeddings/s, the
    query-sharded evaluation with the gallery all-gather), and the evaluation's mAP equal to the single-process value."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CREID_DIST_BACKEND="gloo", CREID_SINGLE_DEVICE="1",
               CREID_BENCH_NO_INSITU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 2 * line["config"]["P"] * line["config"]["K"]
    assert line["value"] > 0 and line["higher_is_better"] is True and line["unit"] == "images/s"
    emb = line.get("embed_ranks") or line["embed"]
    assert emb["config"]["parallelism"] == "dp2" and emb["value"] > 0 and emb["scaling"] == "weak"
    ev = line["eval"]
    assert ev["config"]["parallelism"] == "query-shard x2, gallery all-gather" and ev["value"] > 0
    # the sharded evaluation must reproduce the single-process metric on the same seeded features
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "eval", "--steps", "2", "--warmup", "1",
                             "--no-cpu-baseline"], cwd=ROOT, env={k: v for k, v in env.items() if not k.startswith("CREID_DIST") and k != "CREID_SINGLE_DEVICE"},
                            capture_output=True, text=True, timeout=900)
    assert single.returncode == 0, (single.stdout + single.stderr)[-3000:]
    sline = json.loads([ln for ln in single.stdout.strip().splitlines() if ln.strip()][-1])
    assert abs(ev["mAP"] - sline["mAP"]) < 1e-12, (ev["mAP"], sline["mAP"])


def test_bench_gpus_8_contract_eight_ranks_one_gpu():
    """`bench.py --gpus 8` exactly as the scaling driver will launch it the day an 8-GPU node exists (BASELINE configs[2]),
    with the eight ranks sharing cuda:0 and gloo carrying the collectives: no deadlock in the 3-bucket overlapped gradient
    schedule at world 8 (every rank must agree on the capture fallback before choosing its collective schedule), rc 0, ONE
    stdout line, global batch 8 x P x K (the PK shards of datasets/samplers/distributed_pids_sampler.py:61-71), the
    query-sharded evaluation over the all-gathered gallery."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CREID_DIST_BACKEND="gloo", CREID_SINGLE_DEVICE="1",
               CREID_BENCH_NO_INSITU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "dp8" and line["config"]["global_batch"] == 8 * line["config"]["P"] * line["config"]["K"]
    assert line["value"] > 0 and np.isfinite(line["final_loss"])
    emb = line.get("embed_ranks") or line["embed"]
    assert emb["config"]["parallelism"] == "dp8" and emb["value"] > 0
    assert line["eval"]["config"]["parallelism"] == "query-shard x8, gallery all-gather" and 0 < line["eval"]["mAP"] < 1


def test_f16_training_data_parallel_two_ranks_one_gpu():
    """f16 training (the reference's precision=16) on the overlapped data-parallel path: two gloo ranks on one GPU, the step as
    hipGraph segments with the bucketed gradient all-reduce between them, the loss-scale kernels inside the optimiser segment.
    The ranks need no collective of their own for the overflow decision (a non-finite gradient reaches every rank through the
    SUM all-reduce before the unscale + check runs): the run must finish, with a finite loss and every step either applied or
    skipped on the scale's way down from 65536."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CREID_DIST_BACKEND="gloo", CREID_SINGLE_DEVICE="1",
               CREID_BENCH_NO_INSITU="1", CREID_BENCH_NO_EVAL="1", CREID_BENCH_DTYPE="f16")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["dtype"] == "f16" and line["n_gpus"] == 2 and np.isfinite(line["final_loss"])
    st = line["f16_state"]
    assert 1.0 <= st["loss_scale"] <= 65536.0 and st["adam_steps_applied"] >= 1


def test_lonely_identity_error_is_raised_on_every_rank_two_ranks_one_gpu():
    """with a device isReal mask the lonely-identity error is raised late, from check_lonely_identities(); under data
    parallelism only the rank that saw the batch has a non-zero counter, so the counter is all-reduced first and EVERY rank raises
    (a rank raising alone would leave the others hanging in their next collective)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "debug", "lonely_ddp_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("LONELY_DDP_OK") == 2 and "LONELY_DDP_MISMATCH" not in out, out[-3000:]
