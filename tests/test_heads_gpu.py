"""GPU parity: stages B/C (centroids, triplet, center, xent, BNNeck, classifier, optimiser steps)
through the C ABI against golden vectors from the reference and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from centroids_reid_amd import losses, ops
    return losses, ops


@pytest.mark.parametrize("name", ["losses_n64_d128", "losses_n32_d2048"])
def test_losses_golden(golden, mods, name):
    losses, ops = mods
    g = golden(name)
    x = torch.from_numpy(g["x"]).cuda(); labels = torch.from_numpy(g["labels"]).cuda()
    dm, dap, dan, pi, ni = ops.pairwise_dist_mine(x, labels)
    off = ~np.eye(len(g["x"]), dtype=bool)
    np.testing.assert_allclose(dm.cpu().numpy()[off], g["dist"][off], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(pi.cpu().numpy(), g["p_inds"])
    np.testing.assert_array_equal(ni.cpu().numpy(), g["n_inds"])
    for tag, margin, m in (("m05", 0.5, None), ("soft", None, None), ("m05_mask", 0.5, g["mask"])):
        xt = x.clone().requires_grad_(True)
        mask = None if m is None else torch.from_numpy(m).cuda()
        loss, ap, an = losses.TripletLoss(margin)(xt, labels, mask=mask)
        loss.backward()
        assert abs(loss.item() - float(g[f"trip_{tag}_loss"])) < 1e-5
        np.testing.assert_allclose(ap.cpu().numpy(), g[f"trip_{tag}_ap"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(an.cpu().numpy(), g[f"trip_{tag}_an"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), g[f"trip_{tag}_grad"], rtol=1e-4, atol=1e-6)
    C, D = g["centers"].shape
    cl = losses.CenterLoss(C, D).cuda()
    with torch.no_grad():
        cl.centers.copy_(torch.from_numpy(g["centers"]))
    xt = x.clone().requires_grad_(True)
    l = cl(xt, labels); l.backward()
    assert abs(l.item() - float(g["center_loss"])) < 1e-5 * abs(float(g["center_loss"]))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g["center_grad_x"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(cl.centers.grad.cpu().numpy(), g["center_grad_c"], rtol=1e-4, atol=1e-6)
    lt = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    l = losses.CrossEntropyLabelSmooth(lt.shape[1])(lt, labels); l.backward()
    assert abs(l.item() - float(g["xent_loss"])) < 1e-5
    np.testing.assert_allclose(lt.grad.cpu().numpy(), g["xent_grad"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("P,K,D,fakes", [(16, 4, 128, ()), (16, 4, 2048, (5,)), (8, 4, 64, (8, 9, 30)), (4, 2, 32, (1,)),
                                          (6, 3, 36, (0, 1, 2))])
def test_loo_centroids_vs_oracle(mods, P, K, D, fakes):
    from oracle import reid_oracle as ro
    _, ops = mods
    rng = np.random.default_rng(P * K + D)
    f = torch.from_numpy(rng.standard_normal((P * K, D)).astype(np.float32))
    ir = torch.ones(P * K, dtype=torch.bool)
    for j in fakes:
        ir[j] = False
    ft = f.clone().requires_grad_(True)
    co, vo = ro.loo_centroids(ft, ir, P, K)
    w = torch.from_numpy(rng.standard_normal(tuple(co.shape)).astype(np.float32))
    (co * w).sum().backward()
    fg = f.cuda().requires_grad_(True)
    cg, vg = ops.LooCentroids.apply(fg, ir.cuda(), P, K)
    (cg * w.cuda()).sum().backward()
    np.testing.assert_array_equal(cg.detach().cpu().numpy(), co.detach().numpy())   # same s-order -> bit-exact
    np.testing.assert_array_equal(vg.cpu().numpy(), vo.numpy())
    np.testing.assert_allclose(fg.grad.cpu().numpy(), ft.grad.numpy(), rtol=1e-6, atol=1e-7)


def test_bnneck_and_classifier_vs_oracle(mods):
    losses, ops = mods
    rng = np.random.default_rng(2)
    B, D, C = 61, 2048, 751
    x = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32) * 2 + 0.5)
    w = torch.from_numpy((1 + 0.1 * rng.standard_normal(D)).astype(np.float32))
    fc = torch.from_numpy((rng.standard_normal((C, D)) * 0.01).astype(np.float32))
    coef = torch.from_numpy(rng.standard_normal((B, C)).astype(np.float32))
    # oracle
    xt, wt, fct = x.clone().requires_grad_(True), w.clone().requires_grad_(True), fc.clone().requires_grad_(True)
    rm, rv = torch.zeros(D), torch.ones(D)
    y = torch.nn.functional.batch_norm(xt, rm, rv, wt, torch.zeros(D), True, 0.1, 1e-5)
    (y @ fct.t() * coef).sum().backward()
    # HIP
    bn = losses.BatchNorm1d(D).cuda(); lin = losses.Linear(D, C).cuda()
    with torch.no_grad():
        bn.weight.copy_(w); lin.weight.copy_(fc)
    xg = x.cuda().requires_grad_(True)
    yg = bn(xg)
    lg = lin(yg)
    (lg * coef.cuda()).sum().backward()
    np.testing.assert_allclose(yg.detach().cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), (y @ fct.t()).detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), rv.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), fct.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xt.grad.numpy(), rtol=1e-3, atol=1e-5)
    # eval mode
    bn.eval()
    ye = bn(x.cuda())
    yo = torch.nn.functional.batch_norm(x, rm, rv, w, torch.zeros(D), False, 0.1, 1e-5)
    np.testing.assert_allclose(ye.detach().cpu().numpy(), yo.detach().numpy(), rtol=1e-5, atol=1e-5)


def test_optimizer_steps_vs_oracle(mods):
    from oracle import reid_oracle as ro
    from centroids_reid_amd import _lib as L
    rng = np.random.default_rng(4)
    n = 100003
    p = torch.from_numpy(rng.standard_normal(n).astype(np.float32)); g = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    m = torch.zeros(n); v = torch.zeros(n)
    pg, gg, mg, vg = p.cuda(), g.cuda(), m.cuda(), v.cuda()
    for step in (1, 2, 3):
        p, m, v = ro.adam_step(p, g, m, v, step, 3.5e-5)
        L.check(L.lib().creid_adam_step(L.ptr(pg), L.ptr(gg), L.ptr(mg), L.ptr(vg), n, 3.5e-5, 0.9, 0.999, 1e-8, 5e-4,
                                        step, 1.0, L.stream()), "adam")
    np.testing.assert_allclose(pg.cpu().numpy(), p.numpy(), rtol=1e-5, atol=1e-7)
    # device-resident hyper-parameter variant (hipGraph-safe) gives the same trajectory
    n4 = 100004
    p2 = torch.from_numpy(rng.standard_normal(n4).astype(np.float32)); g2 = torch.from_numpy(rng.standard_normal(n4).astype(np.float32))
    pa, ga, ma, va = p2.cuda(), g2.cuda(), torch.zeros(n4).cuda(), torch.zeros(n4).cuda()
    hyper = torch.tensor([3.5e-5, 0, 0, 0, 0, 0, 0, 0], dtype=torch.float32).cuda()      # float[8] (creid.h)
    pr, mr, vr = p2.clone(), torch.zeros(n4), torch.zeros(n4)
    for step in (1, 2, 3):
        pr, mr, vr = ro.adam_step(pr, g2, mr, vr, step, 3.5e-5)
        L.check(L.lib().creid_adam_step_dev(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), n4, L.ptr(hyper), 0.9, 0.999, 1e-8,
                                            5e-4, 1.0, L.stream()), "adam_dev")
    np.testing.assert_allclose(pa.cpu().numpy(), pr.numpy(), rtol=1e-5, atol=1e-7)
    assert int(hyper[1].item()) == 3
    c = torch.from_numpy(rng.standard_normal(5000).astype(np.float32)); gc = torch.from_numpy(rng.standard_normal(5000).astype(np.float32) * 1e-4)
    cg, gcg = c.cuda(), gc.cuda()
    L.check(L.lib().creid_sgd_scaled_step(L.ptr(cg), L.ptr(gcg), 5000, 0.5, 1.0 / 5e-4, L.stream()), "sgd")
    np.testing.assert_allclose(cg.cpu().numpy(), ro.center_sgd_step(c, gc).numpy(), rtol=1e-6, atol=1e-7)


def test_triplet_cosine_and_normalize_golden(golden, mods):
    """SOLVER.DISTANCE_FUNC='cosine' and normalize_feature=True (losses/triplet_loss.py:44-65,134-143) through the
    HIP kernels vs vectors recorded from the reference."""
    losses, ops = mods
    g = golden("surface_r2")
    x = torch.from_numpy(g["x"]).cuda(); labels = torch.from_numpy(g["labels"]).cuda()
    for tag, margin, dist, norm, m in (("cos_m05", 0.5, "cosine", False, None), ("cos_soft", None, "cosine", False, None),
                                       ("cos_m05_mask", 0.5, "cosine", False, g["mask"]),
                                       ("euc_norm", 0.5, "euclidean", True, None), ("cos_norm", 0.3, "cosine", True, None)):
        xt = x.clone().requires_grad_(True)
        mask = None if m is None else torch.from_numpy(m).cuda()
        loss, ap, an = losses.TripletLoss(margin, dist)(xt, labels, normalize_feature=norm, mask=mask)
        loss.backward()
        assert abs(loss.item() - float(g[f"trip_{tag}_loss"])) < 2e-6, tag
        np.testing.assert_allclose(ap.cpu().numpy(), g[f"trip_{tag}_ap"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(an.cpu().numpy(), g[f"trip_{tag}_an"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), g[f"trip_{tag}_grad"], rtol=2e-4, atol=2e-7)


def test_general_distances_and_mining_golden(golden, mods):
    """euclidean_dist(x, y) / cosine_dist(x, y) for two different row sets (values + both gradients) and
    hard_example_mining(dist_mat, labels, return_inds=True) on a given matrix (losses/triplet_loss.py:27-119)."""
    losses, ops = mods
    g = golden("surface_r2")
    w = torch.from_numpy(g["w"]).cuda()
    for name, fn in (("euc", losses.euclidean_dist), ("cos", losses.cosine_dist)):
        xt = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
        yt = torch.from_numpy(g["y"]).cuda().requires_grad_(True)
        d = fn(xt, yt)
        (d * w).sum().backward()
        np.testing.assert_allclose(d.detach().cpu().numpy(), g[f"xy_{name}_dist"], rtol=1e-5, atol=2e-5 if name == "euc" else 2e-6)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), g[f"xy_{name}_gx"], rtol=2e-4, atol=2e-5 if name == "euc" else 2e-6)
        np.testing.assert_allclose(yt.grad.cpu().numpy(), g[f"xy_{name}_gy"], rtol=2e-4, atol=2e-5 if name == "euc" else 2e-6)
    labels = torch.from_numpy(g["labels"]).cuda()
    dm = torch.from_numpy(g["mine_dist"]).cuda().requires_grad_(True)
    ap, an, pi, ni = losses.hard_example_mining(dm, labels, return_inds=True)
    np.testing.assert_array_equal(pi.cpu().numpy(), g["mine_pi"]); np.testing.assert_array_equal(ni.cpu().numpy(), g["mine_ni"])
    np.testing.assert_array_equal(ap.detach().cpu().numpy(), g["mine_ap"])
    np.testing.assert_array_equal(an.detach().cpu().numpy(), g["mine_an"])
    (ap.sum() - 2 * an.sum()).backward()
    ref = np.zeros_like(g["mine_dist"])
    ref[np.arange(len(ref)), g["mine_pi"]] += 1.0
    ref[np.arange(len(ref)), g["mine_ni"]] -= 2.0
    np.testing.assert_array_equal(dm.grad.cpu().numpy(), ref)
    ap2, an2 = losses.hard_example_mining(dm.detach(), labels)
    assert torch.equal(ap2, ap.detach()) and torch.equal(an2, an.detach())


@pytest.mark.parametrize("P,K,D,fakes", [(16, 4, 2048, ()), (8, 4, 64, (8, 9, 30)), (6, 3, 36, (0, 1, 2))])
def test_loo_emb_kernels_equal_centroid_kernels_plus_glue(P, K, D, fakes):
    """creid_loo_emb_fwd / _bwd = creid_loo_centroids_fwd / _bwd + the strided copies, cat and add_ the training step used to
    do with torch ops; creid_ctl_step_stats = the step's scalar bookkeeping (train_ctl_model.py:143-177)."""
    from centroids_reid_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(P + K + D)
    f32 = dict(dtype=torch.float32, device="cuda")
    feat = torch.from_numpy(rng.standard_normal((P * K, D)).astype(np.float32)).cuda()
    real = torch.ones(P * K, dtype=torch.uint8, device="cuda")
    for j in fakes:
        real[j] = 0
    labels = torch.arange(P, device="cuda").repeat_interleave(K) * 7 + 3
    cent0 = torch.empty((K, P, D), **f32); valid0 = torch.empty((K, P), dtype=torch.int32, device="cuda")
    L.check(lib.creid_loo_centroids_fwd(L.ptr(feat), L.ptr(real), P, K, D, L.ptr(cent0), L.ptr(valid0), L.stream()), "fwd0")
    cent = torch.empty_like(cent0); valid = torch.empty_like(valid0)
    emb = torch.empty((K, 2 * P, D), **f32); lab = torch.empty((K, 2 * P), dtype=torch.int64, device="cuda")
    cnorm = torch.empty(K * P, **f32)
    L.check(lib.creid_loo_emb_fwd(L.ptr(feat), L.ptr(real), L.ptr(labels), P, K, D, L.ptr(cent), L.ptr(valid), L.ptr(emb), L.ptr(lab),
                                  L.ptr(cnorm), L.stream()), "fwd")
    assert torch.equal(cent, cent0) and torch.equal(valid, valid0)
    assert torch.equal(emb[:, :P], feat.view(P, K, D).transpose(0, 1)) and torch.equal(emb[:, P:], cent0)
    lt = labels.view(P, K).t()
    assert torch.equal(lab, torch.cat((lt, lt), dim=1))
    np.testing.assert_allclose(cnorm.cpu().numpy(), torch.linalg.vector_norm(cent0, dim=2).flatten().cpu().numpy(), rtol=2e-6, atol=1e-7)
    # backward: dfeat += query rows of demb, then the leave-one-out adjoint of its centroid rows
    demb = torch.from_numpy(rng.standard_normal((K, 2 * P, D)).astype(np.float32)).cuda()
    base = torch.from_numpy(rng.standard_normal((P * K, D)).astype(np.float32)).cuda()
    ref = base.clone()
    ref.view(P, K, D).add_(demb[:, :P].transpose(0, 1))
    dcent = demb[:, P:].contiguous()
    L.check(lib.creid_loo_centroids_bwd(L.ptr(dcent), L.ptr(real), P, K, D, L.ptr(ref), L.stream()), "bwd0")
    got = base.clone()
    L.check(lib.creid_loo_emb_bwd(L.ptr(demb), L.ptr(real), P, K, D, L.ptr(got), L.stream()), "bwd")
    assert torch.equal(got, ref)
    # scalars
    n = 4 * (K + 1) + 2
    scal = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    w = torch.zeros(n, **f32); w[0] = 1.0; w[4:4 * (K + 1):4] = 1.0 / K; w[4 * (K + 1)] = 5e-4; w[4 * (K + 1) + 1] = 1.0
    out = torch.empty(n + 7, **f32)
    L.check(lib.creid_ctl_step_stats(L.ptr(scal), L.ptr(w), n, K, L.ptr(cnorm), K * P, L.ptr(out), L.stream()), "stats")
    terms = scal * w
    np.testing.assert_array_equal(out[:n].cpu().numpy(), terms.cpu().numpy())
    want = torch.cat([terms.sum().view(1), terms[4:4 * (K + 1):4].sum().view(1), scal[4:4 * (K + 1)].view(K, 4).mean(0), cnorm.mean().view(1)])
    np.testing.assert_allclose(out[n:].cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=1e-6)
