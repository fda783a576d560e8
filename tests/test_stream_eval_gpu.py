"""GPU parity: metric-only evaluation that never writes the m x n matrix (csrc/stream_eval.hip) against the
materialised path (creid_sqdist_matrix + creid_rank_rows + creid_cmc_ap_ranked), the reference goldens and the CPU
oracle -- including the per-rank shard of BASELINE configs[3] (6250 queries x 200 000 gallery x 2048)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(feats, pids, cams, nq, feat_norm=True):
    from centroids_reid_amd import reid_metric as rm
    f = feats.cuda()
    a = rm.R1_mAP(num_query=nq, feat_norm=feat_norm)
    ra = a.compute(f, pids, cams)
    b = rm.R1_mAP(num_query=nq, feat_norm=feat_norm, streamed=True)
    rb = b.compute(f, pids, cams)
    return a, ra, b, rb


def _per_query_from_indices(metric, pids, cams, nq):
    from centroids_reid_amd import reid_metric as rm
    _, _, _, _, valid, ap, first = rm.eval_func_device(metric.last["indices"], pids[:nq], pids[nq:], cams[:nq], cams[nq:], 50)
    return valid.cpu().numpy(), ap.cpu().numpy(), first.cpu().numpy()


def _assert_same(a, ra, b, rb, pids, cams, nq):
    v0, ap0, f0 = _per_query_from_indices(a, pids, cams, nq)
    v1, ap1, f1 = b.last["valid"].cpu().numpy(), b.last["ap"].cpu().numpy(), b.last["first"].cpu().numpy()
    np.testing.assert_array_equal(v1, v0)                       # bit-exact: same ranks on the same distance bits
    np.testing.assert_array_equal(f1, f0)
    np.testing.assert_allclose(ap1, ap0, rtol=0, atol=1e-12)    # float64 sums in a different order
    np.testing.assert_array_equal(rb[0], ra[0])                 # CMC curve
    assert abs(rb[1] - ra[1]) < 1e-12
    np.testing.assert_array_equal(rb[2], ra[2])
    np.testing.assert_allclose(b.last["single_performance"], a.last["single_performance"], rtol=0, atol=1e-12)


@pytest.fixture(params=["0", "1"], ids=["split-major", "equal-runs"])
def work_split(monkeypatch, request):
    """Both work splits of the counting contraction (csrc/stream_eval.hip: mode 0 = per-row slices for galleries beyond the
    Infinity Cache, mode 1 = equal runs of 64-column units that may cross query tiles); the default picks by gallery size."""
    monkeypatch.setenv("CREID_STREAM_BALANCE", request.param)
    return request.param


@pytest.mark.parametrize("name", ["eval_small", "eval_d2048", "eval_tiny_gallery"])
def test_streamed_matches_reference_goldens(golden, name, work_split):
    g = golden(name)
    nq = int(g["num_query"])
    feats = torch.from_numpy(g["feats"])
    a, ra, b, rb = _both(feats, g["pids"], g["camids"], nq)
    np.testing.assert_array_equal(a.last["indices"].cpu().numpy(), g["indices"])     # the materialised path is pinned
    _assert_same(a, ra, b, rb, g["pids"], g["camids"], nq)
    np.testing.assert_allclose(rb[0], g["cmc"], rtol=0, atol=1e-7)
    assert abs(rb[1] - float(g["mAP"])) < 1e-9
    np.testing.assert_allclose(rb[2], g["topk"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("nq,ng,D,npid,ncam,dup", [(300, 3000, 256, 60, 3, True), (70, 513, 100, 9, 2, False),
                                                    (129, 1000, 2048, 400, 5, True), (5, 40, 32, 3, 2, False),
                                                    (33, 300, 8, 5, 2, False), (33, 700, 20, 7, 3, True), (64, 512, 48, 9, 2, False)])
def test_streamed_equals_materialised_random(nq, ng, D, npid, ncam, dup, work_split):
    """N(0,1) features (dense near-ties in fp32), duplicated gallery rows (exact ties -> order by gallery index),
    queries whose pid is absent from the gallery or whose positives all share their camera."""
    rng = np.random.default_rng(nq * 7 + ng)
    f = rng.standard_normal((nq + ng, D)).astype(np.float32)
    pids = rng.integers(0, npid, nq + ng)
    cams = rng.integers(0, ncam, nq + ng)
    if dup:
        src = rng.integers(nq, nq + ng, ng // 4); dst = rng.integers(nq, nq + ng, ng // 4)
        f[dst] = f[src]                                         # exact ties, possibly between a positive and a negative
        f[nq + 7] = f[3]; pids[nq + 7] = pids[3]; cams[nq + 7] = cams[3] + 1      # a zero-distance positive
    pids[0] = npid + 5                                          # pid absent from the gallery
    same = (pids[nq:] == pids[1])
    cams[nq:][same] = cams[1]                                   # every same-pid entry removed -> invalid query
    for norm in (True, False):
        a, ra, b, rb = _both(torch.from_numpy(f), pids, cams, nq, feat_norm=norm)
        _assert_same(a, ra, b, rb, pids, cams, nq)
        assert b.last["valid"][0].item() == 0 and b.last["valid"][1].item() == 0


def test_streamed_overflow_rows_take_general_path(work_split):
    """A pid with more than 128 positives does not fit the LDS list: those queries are routed through the
    materialised kernels and merged; the rest stay streamed."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(5)
    nq, ng, D = 40, 2000, 64
    f = rng.standard_normal((nq + ng, D)).astype(np.float32)
    pids = rng.integers(2, 30, nq + ng)
    pids[nq:nq + 400] = 0; pids[:6] = 0                         # 400 gallery entries of pid 0
    pids[nq + 400:nq + 500] = 1; pids[6:9] = 1                  # 100 of pid 1 (fits: cap 128)
    cams = rng.integers(0, 4, nq + ng)
    a, ra, b, rb = _both(torch.from_numpy(f), pids, cams, nq)
    plan = b.last["plan"]
    assert set(plan.overflow.tolist()) == set(range(6)) and plan.cap == 128
    _assert_same(a, ra, b, rb, pids, cams, nq)


def test_streamed_duke_shape_equals_materialised(work_split):
    """BASELINE configs[4] shape (2228 x 17661 x 2048, the bench generator): whole-job equality of the two paths."""
    gen = torch.Generator(device="cuda").manual_seed(0)
    nq, ng, D = 2228, 17661, 2048
    feats = torch.randn((nq + ng, D), generator=gen, device="cuda", dtype=torch.float32)
    rng = np.random.default_rng(0)
    pids = rng.integers(0, 702, nq + ng); cams = rng.integers(0, 8, nq + ng)
    a, ra, b, rb = _both(feats, pids, cams, nq)
    _assert_same(a, ra, b, rb, pids, cams, nq)
    assert 0 < rb[1] < 1


def test_north_star_3000x15000_streamed_materialised_oracle(work_split):
    """north_star's own target shape (3000 queries x 15000 gallery x 2048 fp32; utils/reid_metric.py:112-151): the streamed
    path equals the materialised one, the ranking has the size-independent properties, the integer stage equals the oracle on
    the device's ranking and a 64-query slice of the distance matrix matches float64 CPU arithmetic."""
    from oracle import reid_oracle as ro
    gen = torch.Generator(device="cuda").manual_seed(3000)
    nq, ng, D = 3000, 15000, 2048
    feats = torch.randn((nq + ng, D), generator=gen, device="cuda", dtype=torch.float32)
    rng = np.random.default_rng(3000)
    pids = rng.integers(0, 751, nq + ng); cams = rng.integers(0, 6, nq + ng)
    a, ra, b, rb = _both(feats, pids, cams, nq)
    _assert_same(a, ra, b, rb, pids, cams, nq)
    d, idx = a.last["distmat"], a.last["indices"]
    srt = torch.sort(idx, dim=1).values
    assert torch.equal(srt, torch.arange(ng, device="cuda").expand(nq, ng))          # every row a permutation
    ds = torch.gather(d, 1, idx)
    assert bool((ds[:, 1:] >= ds[:, :-1]).all())                                     # non-decreasing distances
    tie = ds[:, 1:] == ds[:, :-1]
    assert bool((idx[:, 1:][tie] > idx[:, :-1][tie]).all())                          # ties in gallery-index order
    assert float(d.min()) > 0.0 and float(d.max()) < 4.0                             # unit vectors
    cmc_o, mAP_o, topk_o, _ = ro.eval_market(idx.cpu().numpy(), pids[:nq], pids[nq:], cams[:nq], cams[nq:])
    np.testing.assert_array_equal(ra[0], cmc_o)
    assert abs(ra[1] - mAP_o) < 1e-12 and abs(rb[1] - mAP_o) < 1e-12
    qn, gn = ro.l2_normalize(feats[:64].cpu().double()), ro.l2_normalize(feats[nq:nq + 512].cpu().double())
    np.testing.assert_allclose(d[:64, :512].cpu().numpy(), ro.sqdist_matrix(qn, gn).numpy(), rtol=0, atol=5e-6)


def test_streamed_configs3_shard_6250x200000():
    """The per-rank shard of BASELINE configs[3]: 6250 queries x 200 000 gallery x 2048 fp32, streamed (the 5 GB
    distance matrix and the 10 GB index matrix are never written).  Full-size properties + equality with the
    materialised kernels AND the CPU oracle's CMC/AP on a 256-query slice."""
    from centroids_reid_amd import reid_metric as rm
    from oracle import reid_oracle as ro
    nq, ng, D, npid = 6250, 200_000, 2048, 50_000
    gen = torch.Generator(device="cuda").manual_seed(4)
    feats = torch.randn((nq + ng, D), generator=gen, device="cuda", dtype=torch.float32)
    rng = np.random.default_rng(4)
    pids = np.concatenate([rng.integers(0, npid, nq), np.arange(ng) % npid])       # every query has 4 gallery matches
    cams = np.concatenate([np.zeros(nq, np.int64), np.ones(ng, np.int64)])          # datasets/bases.py:226-229
    cams[nq:nq + 5000] = 0                                                           # some same-camera entries to remove
    m = rm.R1_mAP(num_query=nq, streamed=True)
    cmc, mAP, topk = m.compute(feats, pids, cams)
    valid, ap, first = m.last["valid"].cpu().numpy(), m.last["ap"].cpu().numpy(), m.last["first"].cpu().numpy()
    plan = m.last["plan"]
    assert plan.cap == 4 and len(plan.overflow) == 0
    np.testing.assert_array_equal(valid == 1, plan.n_pos > 0)
    assert (first[valid == 1] >= 0).all() and (first[valid == 1] < ng).all()
    assert ((ap[valid == 1] > 0) & (ap[valid == 1] <= 1)).all()
    assert np.all(np.diff(cmc) >= 0) and 0 < mAP < 1
    assert abs(mAP - ap[valid == 1].mean()) < 1e-12
    # 256-query slice through the materialised kernels and through the oracle's eval on the device's own distances
    sl = np.arange(1000, 1256)
    fn, sq = rm.l2_normalize(feats, return_sqnorm=True)
    d = rm.get_euclidean(fn[:nq][sl[0]:sl[-1] + 1], fn[nq:], sq[:nq][sl[0]:sl[-1] + 1].contiguous(), sq[nq:].contiguous())
    idx = rm.rank_rows(d)
    _, _, _, _, v2, a2, f2 = rm.eval_func_device(idx, pids[:nq][sl], pids[nq:], cams[:nq][sl], cams[nq:], 50)
    np.testing.assert_array_equal(valid[sl], v2.cpu().numpy())
    np.testing.assert_array_equal(first[sl], f2.cpu().numpy())
    np.testing.assert_allclose(ap[sl], a2.cpu().numpy(), rtol=0, atol=1e-12)
    o_idx = ro.rank_rows(d.cpu())
    _, _, _, per_q = ro.eval_market(o_idx, pids[:nq][sl], pids[nq:], cams[:nq][sl], cams[nq:])
    np.testing.assert_array_equal(valid[sl] == 1, per_q["valid"])
    np.testing.assert_array_equal(first[sl], per_q["first"])
    np.testing.assert_allclose(ap[sl], per_q["ap"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("nq,ng,npid,ncam,offset", [(50, 700, 40, 4, 0), (2228, 17661, 702, 8, 0), (300, 5000, 2000, 3, -977),
                                                    (64, 3000, 5, 2, 10_000_000)])
def test_device_stream_plan_equals_host_plan(nq, ng, npid, ncam, offset):
    """creid_stream_plan (counting sort on the device) against the numpy construction: same capacity, overflow set and
    per-query positive counts, and for every query the same GROUP of gallery indices (the order inside a group is free)."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(nq + ng)
    pids = rng.integers(0, npid, nq + ng) + offset
    cams = rng.integers(0, ncam, nq + ng)
    pids[0] = offset + npid + 3                                  # above the gallery's range
    pids[1] = offset - 2                                         # below it
    if npid > 10:
        pids[nq:][pids[nq:] == offset + 7] = offset + 8          # a hole inside the range
        pids[2] = offset + 7
    dev = rm.StreamPlan.on_device(pids, cams, nq, "cuda").finish()
    host = rm.StreamPlan(pids[:nq], pids[nq:], cams[:nq], cams[nq:], "cuda")
    assert dev.cap == host.cap
    np.testing.assert_array_equal(dev.overflow, host.overflow)
    np.testing.assert_array_equal(dev.n_pos, host.n_pos)
    ds, hs = dev.q_slot[:nq].cpu().numpy(), host.q_slot.cpu().numpy()
    np.testing.assert_array_equal(ds < 0, hs < 0)
    dc, hc = dev.csr_off.cpu().numpy(), host.csr_off.cpu().numpy()
    do, ho = dev.g_order.cpu().numpy(), host.g_order.cpu().numpy()
    assert sorted(do.tolist()) == list(range(ng))                # a permutation of the gallery
    for qi in range(0, nq, max(1, nq // 97)):
        if hs[qi] < 0:
            continue
        a = np.sort(do[dc[ds[qi]]:dc[ds[qi] + 1]]); b = np.sort(ho[hc[hs[qi]]:hc[hs[qi] + 1]])
        np.testing.assert_array_equal(a, b)
        assert (pids[nq:][a] == pids[qi]).all()


def test_device_plan_sparse_pid_range_falls_back_to_host_index():
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(3)
    nq, ng = 20, 400
    pids = rng.integers(0, 30, nq + ng) * (1 << 40)              # range far beyond the dense counting sort
    cams = rng.integers(0, 3, nq + ng)
    f = torch.from_numpy(rng.standard_normal((nq + ng, 64)).astype(np.float32))
    a, ra, b, rb = _both(f, pids, cams, nq)
    _assert_same(a, ra, b, rb, pids, cams, nq)


def test_streamed_speculative_capacity_is_verified():
    """A streamed evaluation of a shape seen before assumes the earlier call's positive-list capacity instead of reading the
    plan's statistics back in the middle of the pipeline, and checks the assumption in the final read-back: right hint ->
    same results as the synchronous first call; too small -> redone with the right capacity; larger than needed -> identical
    results, the hint shrinks."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(77)
    nq, ng, D = 64, 1500, 96
    f = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32)).cuda()
    few = (rng.integers(0, 300, nq + ng), rng.integers(0, 3, nq + ng))          # ~5 gallery entries per identity
    many = (rng.integers(0, 20, nq + ng), rng.integers(0, 3, nq + ng))          # ~75 per identity (< 128: no overflow rows)

    def run(labels):
        m = rm.R1_mAP(num_query=nq, streamed=True)
        r = m.compute(f, labels[0], labels[1])
        return m, r

    def same(r0, m0, r1, m1):
        np.testing.assert_array_equal(r1[0], r0[0]); assert r1[1] == r0[1]; np.testing.assert_array_equal(r1[2], r0[2])
        for k in ("valid", "ap", "first"):
            assert torch.equal(m1.last[k], m0.last[k]), k

    rm._CAP_HINT.clear()
    m0, r0 = run(few)                                     # synchronous: no hint yet
    cap_few = rm._CAP_HINT[(nq, ng)]
    assert cap_few == m0.last["plan"].cap and cap_few <= 32
    m1, r1 = run(few)                                     # speculative, hint right
    same(r0, m0, r1, m1)
    rm._CAP_HINT.clear()
    mb0, rb0 = run(many)                                  # reference for the second label set, synchronous
    cap_many = mb0.last["plan"].cap
    assert cap_many > cap_few
    rm._CAP_HINT[(nq, ng)] = cap_few                      # a hint that is too small
    mb1, rb1 = run(many)
    same(rb0, mb0, rb1, mb1)
    assert mb1.last["plan"].cap == cap_many and rm._CAP_HINT[(nq, ng)] == cap_many
    m2, r2 = run(few)                                     # hint larger than needed: same results, hint shrinks
    assert m2.last["plan"].cap == cap_many
    same(r0, m0, r2, m2)
    assert rm._CAP_HINT[(nq, ng)] == cap_few
    a = rm.R1_mAP(num_query=nq)                           # and the materialised path agrees
    ra = a.compute(f, many[0], many[1])
    np.testing.assert_array_equal(rb1[0], ra[0]); assert abs(rb1[1] - ra[1]) < 1e-12


def test_streamed_results_do_not_depend_on_the_gallery_slicing():
    """The grid rules of the streamed contraction (per-row slices of at most CREID_STREAM_TPER gallery tiles; equal runs of 64-column
    units over CREID_STREAM_WGS workgroups, which may end inside one query tile and continue in the next) only change how the work
    is cut; a query's histogram is summed over the slices with integer atomics, so valid / first / AP must be
    IDENTICAL for every slicing.  The rule is read once per process: one subprocess per value, same seeded inputs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from centroids_reid_amd import reid_metric as rm\n"
        "nq, ng, D = 333, 9000, 512\n"
        "gen = torch.Generator(device='cuda').manual_seed(11)\n"
        "f = torch.randn((nq + ng, D), generator=gen, device='cuda')\n"
        "rng = np.random.default_rng(11)\n"
        "pids = rng.integers(0, 150, nq + ng); cams = rng.integers(0, 5, nq + ng)\n"
        "m = rm.R1_mAP(num_query=nq, streamed=True)\n"
        "cmc, mAP, topk = m.compute(f, pids, cams)\n"
        "h = hashlib.sha256()\n"
        "for k in ('valid', 'ap', 'first'):\n"
        "    h.update(m.last[k].cpu().numpy().tobytes())\n"
        "print('RESULT', repr(mAP), h.hexdigest())\n" % root)
    outs = []
    for extra in ({"CREID_STREAM_BALANCE": "0", "CREID_STREAM_TPER": "1"}, {"CREID_STREAM_BALANCE": "0", "CREID_STREAM_TPER": "3"},
                  {"CREID_STREAM_BALANCE": "0", "CREID_STREAM_TPER": "1000"}, {"CREID_STREAM_BALANCE": "1"},
                  {"CREID_STREAM_BALANCE": "1", "CREID_STREAM_WGS": "37"}, {"CREID_STREAM_BALANCE": "1", "CREID_STREAM_WGS": "4000"}, {}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1])
    assert all(o == outs[0] for o in outs), outs
