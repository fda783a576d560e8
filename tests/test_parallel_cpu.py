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


def _grad_sync_tail_case(rank, world):
    """The centers' gradient as the TAIL of the flat gradient buffer (solver.build_optimizer): one collective carries both,
    the 1 / world of the centers goes into the SGD kernel's scale."""
    import types
    from centroids_reid_amd import parallel as par
    gflat = torch.arange(16, dtype=torch.float32) * (rank + 1)           # 10 Adam gradients + 6 of the centers
    cgrad = gflat[10:].view(2, 3)
    opt = types.SimpleNamespace(gflat=gflat, grad_scale=1.0)
    opt_c = types.SimpleNamespace(grad_in_adam_tail=True, grad_scale=1.0)
    model = types.SimpleNamespace(optimizers=lambda: (opt, opt_c),
                                  center_loss=types.SimpleNamespace(centers=types.SimpleNamespace(grad=cgrad)))
    par.make_grad_sync(world)(model)
    return (opt.gflat * opt.grad_scale).numpy(), (cgrad * opt_c.grad_scale).numpy()


def test_grad_sync_centers_in_tail_gloo():
    for g, c in _run(_grad_sync_tail_case):
        np.testing.assert_allclose(g, np.arange(16) * 1.5)
        np.testing.assert_allclose(c.reshape(-1), np.arange(10, 16) * 1.5)


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


def test_grad_buckets_cover_flat_buffer_in_reverse_layer_order():
    """Bucket ranges: contiguous, disjoint, cover the whole padded buffer, ordered layer4+heads -> layer3 -> rest."""
    from centroids_reid_amd import parallel as par
    names = ["backbone.base.conv1.weight", "backbone.base.bn1.weight", "backbone.base.layer1.0.conv1.weight",
             "backbone.base.layer2.0.conv1.weight", "backbone.base.layer3.0.conv1.weight", "backbone.base.layer3.1.bn1.bias",
             "backbone.base.layer4.0.conv1.weight", "backbone.base.layer4.2.bn3.weight", "bn.weight", "fc_query.weight"]
    numels = [9408, 64, 4096, 32768, 262144, 256, 1048576, 2048, 2048, 1538048]
    offs, o = [], 0
    for n in numels:
        offs.append(o); o += (n + 3) // 4 * 4
    b = par.GradBuckets(names, offs, o)
    assert len(b) == 3
    assert b.ranges[0] == (offs[6], o) and b.ranges[1] == (offs[4], offs[6]) and b.ranges[2] == (0, offs[4])
    # a backbone without the named groups degrades to one bucket
    b1 = par.GradBuckets(["a", "b"], [0, 8], 16)
    assert b1.ranges == [(0, 16)]


def _bucket_case(rank, world):
    from centroids_reid_amd import parallel as par
    names = ["stem.w", "layer1.0.w", "layer3.0.w", "layer3.1.w", "layer4.0.w", "bn.weight", "fc_query.weight"]
    numels = [12, 20, 40, 8, 64, 4, 32]
    offs, o = [], 0
    for n in numels:
        offs.append(o); o += (n + 3) // 4 * 4
    b = par.GradBuckets(names, offs, o)
    g = torch.arange(o, dtype=torch.float32) * (rank + 1) + rank
    flat = g.clone()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    for i in range(len(b)):                       # bucket by bucket, in readiness order
        b.all_reduce(g, i)
    # ragged all-gather through ONE collective
    rows = 3 + 2 * rank
    local = torch.arange(rows * 4, dtype=torch.float32).view(rows, 4) + 100 * rank
    allr = par.all_gather_rows(local)
    eq = par.all_gather_rows(torch.full((2, 3), float(rank)))
    return bool(torch.equal(g, flat)), allr.numpy(), eq.numpy(), len(b)


def test_bucketed_allreduce_equals_flat_and_ragged_allgather_gloo():
    out = _run(_bucket_case)
    expect = np.concatenate([np.arange(3 * 4).reshape(3, 4), np.arange(5 * 4).reshape(5, 4) + 100]).astype(np.float32)
    for same, allr, eq, nb in out:
        assert same and nb == 3
        np.testing.assert_array_equal(allr, expect)
        np.testing.assert_array_equal(eq, np.repeat([[0.0], [1.0]], 2, axis=0).repeat(3, axis=1))


def test_bucket_group_names_parse_to_layer_numbers():
    """parallel._layer_number: "layer4." -> 4 by a real parse (str.strip("layer.") is a character-set strip)."""
    from centroids_reid_amd import parallel
    assert parallel._layer_number("layer4.") == 4 and parallel._layer_number("layer3") == 3
    assert parallel._layer_number("layer12.") == 12
    import pytest as _pt
    with _pt.raises(ValueError):
        parallel._layer_number("stem.")
