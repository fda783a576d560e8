"""GPU parity: stage D/E (normalise -> squared-L2 matrix -> rank -> CMC/mAP) through the C ABI,
against (1) golden vectors from the reference, (2) the CPU oracle on seeded inputs,
(3) size-independent properties at BASELINE.json's full DukeMTMC shape (2228 x 17661 x 2048)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rm():
    assert torch.cuda.is_available()
    from centroids_reid_amd import reid_metric
    return reid_metric


@pytest.mark.parametrize("name", ["eval_small", "eval_d2048", "eval_tiny_gallery"])
def test_golden_bit_exact_rank(golden, rm, name):
    g = golden(name)
    nq = int(g["num_query"])
    metric = rm.R1_mAP(num_query=nq)
    cmc, mAP, topk = metric.compute(torch.from_numpy(g["feats"]).cuda(), g["pids"], g["camids"])
    idx = metric.last["indices"].cpu().numpy()
    # bit-exact CMC rank indices on the gap-designed gallery (BASELINE north_star)
    np.testing.assert_array_equal(idx, g["indices"])
    assert abs(mAP - float(g["mAP"])) < 1e-12                 # same integer ranks -> same float64 AP
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(topk, g["topk"], rtol=0, atol=1e-12)
    if "distmat" in g:
        np.testing.assert_allclose(metric.last["distmat"].cpu().numpy(), g["distmat"], rtol=0, atol=2e-6)
    single = g["single"]
    np.testing.assert_array_equal(metric.last["single_performance"][:, 0], single[:, 0])
    np.testing.assert_allclose(metric.last["single_performance"][:, 2], single[:, 2], rtol=0, atol=1e-12)


def test_normalize_and_sqnorm_vs_oracle(rm):
    from oracle import reid_oracle as ro
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((333, 2048)).astype(np.float32) * 3)
    x[7] = 0  # zero row -> eps path
    y, sq = rm.l2_normalize(x.cuda(), return_sqnorm=True)
    np.testing.assert_allclose(y.cpu().numpy(), ro.l2_normalize(x).numpy(), rtol=0, atol=2e-7)  # fp32 embeddings <= 1e-4
    np.testing.assert_allclose(sq.cpu().numpy(), (ro.l2_normalize(x) ** 2).sum(1).numpy(), rtol=0, atol=1e-6)
    yb = rm.l2_normalize(x.cuda(), out_dtype=torch.bfloat16)
    np.testing.assert_allclose(yb.float().cpu().numpy(), ro.l2_normalize(x).numpy(), rtol=0, atol=2 ** -8)


@pytest.mark.parametrize("m,n,D", [(1, 1, 4), (5, 130, 36), (129, 257, 64), (300, 1000, 2048), (64, 64, 20)])
def test_sqdist_fp32_vs_oracle(rm, m, n, D):
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(m * 7 + n)
    q = torch.from_numpy(rng.standard_normal((m, D)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32))
    d = rm.get_euclidean(q.cuda(), g.cuda()).cpu()
    ref = ro.sqdist_matrix(q.double(), g.double())
    # fp32 accumulation error bound: ~ D * eps * |q||g|
    tol = 4e-7 * D * 1.0 + 1e-5
    np.testing.assert_allclose(d.numpy(), ref.numpy(), rtol=1e-5, atol=tol)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_sqdist_16bit_vs_oracle(rm, dt):
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(5)
    q = torch.from_numpy(rng.standard_normal((200, 512)).astype(np.float32)).to(dt)
    g = torch.from_numpy(rng.standard_normal((777, 512)).astype(np.float32)).to(dt)
    d = rm.get_euclidean(q.cuda(), g.cuda()).cpu()
    ref = ro.sqdist_matrix(q.double(), g.double())   # same rounded inputs, exact arithmetic
    np.testing.assert_allclose(d.numpy(), ref.numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 63), (2, 64), (5, 65), (7, 1023), (4, 1025), (3, 5000), (600, 333),
                                 (2, 21504), (2, 21505), (3, 30000)])   # LDS-bucket limit and beyond (radix kernel)
def test_rank_rows_matches_stable_argsort(rm, m, n):
    rng = np.random.default_rng(n)
    d = rng.standard_normal((m, n)).astype(np.float32)
    d[:, : n // 3] = np.round(d[:, : n // 3], 1)        # many exact ties -> index order decides
    if n > 4:
        d[0, 1] = -0.0; d[0, 3] = 0.0; d[0, 2] = -1e-30
    idx = rm.rank_rows(torch.from_numpy(d).cuda()).cpu().numpy()
    np.testing.assert_array_equal(idx, np.argsort(d + 0.0, axis=1, kind="stable"))


def test_rank_rows_lds_path_and_radix_fallback(rm):
    """n in the LDS-bucket range: ordinary rows, rows with heavy exact ties, all-equal rows (largest bucket over
    the limit -> flagged -> radix kernel), an outlier that stretches the key range, +-inf."""
    rng = np.random.default_rng(77)
    m, n = 37, 6000
    d = (2.0 + 0.1 * rng.standard_normal((m, n))).astype(np.float32)       # unit-vector-like distances
    d[1] = 1.25                                                            # all equal -> fallback
    d[2, : n // 2] = 0.5                                                   # half the row tied -> fallback
    d[3] = np.round(d[3], 2)                                               # many small tie groups -> LDS path
    d[4, 17] = 1e-3; d[4, 18] = 3.9                                        # outliers stretch the bucket range
    d[5, 5] = np.inf; d[5, 6] = -np.inf
    d[6] = np.sort(d[6]); d[7] = np.sort(d[7])[::-1]
    idx = rm.rank_rows(torch.from_numpy(d).cuda()).cpu().numpy()
    np.testing.assert_array_equal(idx, np.argsort(d + 0.0, axis=1, kind="stable"))


def test_eval_func_integer_exact_vs_oracle(rm):
    """CMC/AP scan is integer work: fed the SAME ranked indices it must equal the oracle exactly."""
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(9)
    nq, ng = 257, 3001
    idx = np.stack([rng.permutation(ng) for _ in range(nq)]).astype(np.int64)
    qp = rng.integers(0, 50, nq); gp = rng.integers(0, 48, ng)   # pids 48,49 never in gallery -> invalid queries
    qc = rng.integers(0, 3, nq); gc = rng.integers(0, 3, ng)
    cmc, mAP, topk, single = rm.eval_func(torch.from_numpy(idx).cuda(), qp, gp, qc, gc)
    cmc_o, mAP_o, topk_o, ex = ro.eval_market(idx, qp, gp, qc, gc)
    assert (~ex["valid"]).sum() > 0
    np.testing.assert_array_equal(single[:, 0].astype(np.int64), np.nonzero(ex["valid"])[0])
    np.testing.assert_allclose(single[:, 2], ex["ap"][ex["valid"]], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(cmc, cmc_o)
    assert abs(mAP - mAP_o) < 1e-13
    np.testing.assert_allclose(topk, topk_o, rtol=0, atol=1e-15)


def test_pipeline_vs_oracle_random_features(rm):
    """mAP within 1e-4 of the CPU path; indices equal wherever the oracle's adjacent gap > tau."""
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(17)
    nq, ng, D = 150, 2500, 512
    f = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32))
    pids = rng.integers(0, 120, nq + ng); cams = rng.integers(0, 6, nq + ng)
    metric = rm.R1_mAP(num_query=nq)
    cmc, mAP, topk = metric.compute(f.cuda(), pids, cams)
    cmc_o, mAP_o, topk_o, ex = ro.r1_map(f, pids, cams, nq)
    assert abs(mAP - mAP_o) < 1e-4
    np.testing.assert_allclose(cmc, cmc_o, rtol=0, atol=1e-4 + 1.0 / nq)
    idx = metric.last["indices"].cpu().numpy()
    ds = np.take_along_axis(ex["dist"].numpy(), ex["indices"], 1)
    safe = np.ones_like(idx, bool)
    gap = np.diff(ds, axis=1) > 4e-6
    safe[:, 1:] &= gap; safe[:, :-1] &= gap
    assert safe.mean() > 0.5
    np.testing.assert_array_equal(idx[safe], ex["indices"][safe])


def test_full_size_duke_properties(rm):
    """BASELINE config 5 shape: 2228 x 17661 x 2048 -- size-independent checks."""
    from oracle import reid_oracle as ro
    gen = torch.Generator(device="cuda").manual_seed(0)
    nq, ng, D = 2228, 17661, 2048
    f = torch.randn((nq + ng, D), generator=gen, device="cuda", dtype=torch.float32)
    rng = np.random.default_rng(0)
    pids = rng.integers(0, 702, nq + ng); cams = rng.integers(0, 8, nq + ng)
    metric = rm.R1_mAP(num_query=nq)
    cmc, mAP, topk = metric.compute(f, pids, cams)
    d, idx = metric.last["distmat"], metric.last["indices"]
    # (1) every ranked row is a permutation of 0..ng-1
    srt = torch.sort(idx, dim=1).values
    assert torch.equal(srt, torch.arange(ng, device="cuda").expand(nq, ng))
    # (2) distances non-decreasing along the ranking, ties in index order
    ds = torch.gather(d, 1, idx)
    assert bool((ds[:, 1:] >= ds[:, :-1]).all())
    tie = ds[:, 1:] == ds[:, :-1]
    assert bool((idx[:, 1:][tie] > idx[:, :-1][tie]).all())
    # (3) distance sanity on unit vectors: |q-g|^2 = 2 - 2cos in [0, 4]
    assert float(d.min()) > 0.0 and float(d.max()) < 4.0
    # (4) the integer stage equals the oracle exactly on the device's own ranking
    cmc_o, mAP_o, topk_o, _ = ro.eval_market(idx.cpu().numpy(), pids[:nq], pids[nq:], cams[:nq], cams[nq:])
    np.testing.assert_array_equal(cmc, cmc_o)
    assert abs(mAP - mAP_o) < 1e-12
    # (5) a 64-query slice of the matrix against float64 CPU arithmetic
    fn = ro.l2_normalize(f[:64].cpu().double()), ro.l2_normalize(f[nq:nq + 512].cpu().double())
    np.testing.assert_allclose(d[:64, :512].cpu().numpy(), ro.sqdist_matrix(*fn).numpy(), rtol=0, atol=5e-6)


def test_chunked_compute_equals_full(rm):
    rng = np.random.default_rng(23)
    nq, ng, D = 333, 1500, 128
    f = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32)).cuda()
    pids = rng.integers(0, 60, nq + ng); cams = rng.integers(0, 5, nq + ng)
    cmc, mAP, topk = rm.R1_mAP(num_query=nq).compute(f, pids, cams)
    cmc2, mAP2, topk2 = rm.R1_mAP(num_query=nq).compute_chunked(f, pids, cams, query_chunk=100)
    assert abs(mAP - mAP2) < 1e-12
    np.testing.assert_allclose(cmc, cmc2, rtol=0, atol=1e-7)
    np.testing.assert_allclose(topk, topk2, rtol=0, atol=1e-12)


@pytest.mark.parametrize("m,n,k", [(7, 50, 1), (33, 1000, 15), (64, 17661, 100), (5, 4000, 1024), (3, 200000, 50)])
def test_topk_rows_equals_rank_prefix(m, n, k):
    """creid_topk_rows == the first k columns of the stable rank (ties by gallery index), distances included; rows made
    of duplicated values (exact ties across the k-th position) and a constant row (candidate overflow -> flagged,
    served by the rank kernel) included."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(m * 31 + n)
    d = rng.standard_normal((m, n)).astype(np.float32) * 3
    d[0] = np.round(d[0], 1)                                   # heavy ties, resolved by index
    if m > 2:
        d[1] = 0.25                                             # constant row
        d[2, ::3] = -7.0
    dt = torch.from_numpy(d).cuda()
    idx, dsel = rm.topk_rows(dt, k)
    ref = rm.rank_rows(dt)[:, :k]
    assert torch.equal(idx, ref)
    assert torch.equal(dsel, torch.gather(dt, 1, ref))
    np.testing.assert_array_equal(idx.cpu().numpy(), np.argsort(d, axis=1, kind="stable")[:, :k])


@pytest.mark.parametrize("m,n", [(5, 40), (37, 6000), (129, 17661), (9, 3000), (3, 30000)])
def test_rank_rows_eval_equals_rank_then_eval(rm, m, n):
    """creid_rank_rows_eval (ranked rows evaluated while still in LDS; other rows through the scan kernel) == creid_rank_rows
    followed by creid_cmc_ap_ranked, bit for bit -- the one-pass rank kernel (512 <= n <= 21.5 k), its flagged rows
    (pathological ties -> radix kernel + scan), galleries outside its range, queries without any match -- and both equal to
    the CPU oracle's per-query results on a stable argsort of the same matrix."""
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(m * 13 + n)
    d = (2.0 + 0.1 * rng.standard_normal((m, n))).astype(np.float32)
    if m > 4:
        d[1] = 1.25                                             # all equal -> flagged where the LDS kernel applies
        d[2, : n // 2] = 0.5
        d[3] = np.round(d[3], 2)
    qp = rng.integers(0, 12, m); gp = rng.integers(0, 11, n)    # pid 11 never in the gallery -> invalid queries
    qp[0] = 11
    qc = rng.integers(0, 3, m); gc = rng.integers(0, 3, n)
    dt = torch.from_numpy(d).cuda()
    idx0 = rm.rank_rows(dt)
    _, _, _, _, v0, a0, f0 = rm.eval_func_device(idx0, qp, gp, qc, gc, 50)
    idx1, v1, a1, f1 = rm.rank_rows_eval(dt, qp, gp, qc, gc)
    assert torch.equal(idx1, idx0)
    assert torch.equal(v1, v0) and torch.equal(f1, f0) and torch.equal(a1, a0)
    assert int(v1[0]) == 0
    o_idx = np.argsort(d + 0.0, axis=1, kind="stable")
    np.testing.assert_array_equal(idx1.cpu().numpy(), o_idx)
    _, _, _, ex = ro.eval_market(o_idx, qp, gp, qc, gc)
    np.testing.assert_array_equal(v1.cpu().numpy() == 1, ex["valid"])
    np.testing.assert_array_equal(f1.cpu().numpy()[ex["valid"]], ex["first"][ex["valid"]])
    np.testing.assert_allclose(a1.cpu().numpy()[ex["valid"]], ex["ap"][ex["valid"]], rtol=0, atol=1e-14)
