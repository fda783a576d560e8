"""CPU (gloo, world_size 2): the N>1 paths -- gradient all-reduce(mean) hook, query-sharded evaluation with
embedding all-gather -- give the single-process answers.  Compute on CPU comes from the oracle (test infra)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _grad_sync_case(rank, world):
    import types
    from centroids_reid_amd import parallel as par
    torch.manual_seed(rank)
    gflat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    cgrad = torch.ones(3, 4) * (rank + 1)
    opt = types.SimpleNamespace(gflat=gflat, grad_scale=1.0)
    model = types.SimpleNamespace(optimizers=lambda: (opt, None),
                                  center_loss=types.SimpleNamespace(centers=types.SimpleNamespace(grad=cgrad)))
    par.make_grad_sync(world)(model)
    bufs = [torch.full((5,), float(rank + 1))]
    par.allreduce_mean_(bufs)
    return (opt.gflat * opt.grad_scale).numpy(), cgrad.numpy(), bufs[0].numpy()


def test_grad_sync_mean_gloo():
    out = _run(_grad_sync_case)
    for g, c, b in out:
        np.testing.assert_allclose(g, np.arange(10) * 1.5)      # mean of (1x, 2x)
        np.testing.assert_allclose(c, 1.5)
        np.testing.assert_allclose(b, 1.5)


def _eval_case(rank, world):
    from centroids_reid_amd import parallel as par
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(0)
    nq, ng, D = 37, 301, 32
    feats = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32))
    pids = rng.integers(0, 20, nq + ng); cams = rng.integers(0, 3, nq + ng)
    lo, hi = par.shard_bounds(nq + ng, rank, world)

    def per_query(full, q_lo, q_hi):
        fn = ro.l2_normalize(full)
        d = ro.sqdist_matrix(fn[q_lo:q_hi], fn[nq:])
        idx = ro.rank_rows(d)
        _, _, _, ex = ro.eval_market(idx, pids[q_lo:q_hi], pids[nq:], cams[q_lo:q_hi], cams[nq:]) if q_hi > q_lo else (0, 0, 0, dict(valid=[], ap=[], first=[]))
        return ex["valid"], ex["ap"], ex["first"]

    cmc, mAP, topk = par.evaluate_sharded(feats[lo:hi].contiguous(), pids, cams, nq, per_query)
    cmc0, mAP0, topk0, _ = ro.r1_map(feats, pids, cams, nq)
    return float(abs(mAP - mAP0)), float(np.abs(cmc - cmc0).max()), float(np.abs(topk - topk0).max())


def test_sharded_eval_matches_single_process_gloo():
    for dm, dc, dt in _run(_eval_case):
        assert dm < 1e-12 and dc < 1e-7 and dt < 1e-12


def test_shard_bounds_cover():
    from centroids_reid_amd import parallel as par
    for n in (0, 1, 7, 64, 2228):
        for w in (1, 2, 3, 8):
            b = [par.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert [hi - lo for lo, hi in b] == [len(a) for a in np.array_split(np.arange(n), w)]
