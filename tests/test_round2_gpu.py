"""GPU: round-2 fixes to the host side of the hot path -- optimiser state restore, stale-weight detection in the
backbone engine, train-mode restore after validation, untruncated top-k hits, camera-id validation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(K=4, D=2048):
    from centroids_reid_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.MODEL.PRETRAINED = False
    cfg.DATALOADER.NUM_INSTANCE = K
    cfg.USE_MIXED_PRECISION = False
    return cfg


def test_fused_adam_matches_torch_and_restores_state():
    """FusedAdam == torch.optim.Adam over several steps (parameter sizes NOT multiples of 4: the flat buffer pads
    every tensor to 16 bytes), and state_dict -> load_state_dict -> step continues exactly (moments AND the
    bias-correction step count), both from its own state and from a torch.optim.Adam state."""
    from centroids_reid_amd.solver import FusedAdam
    rng = np.random.default_rng(0)
    shapes = [(7, 3), (5,), (16, 4), (3, 3, 3), (1,)]
    init = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in shapes]
    grads = [[torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in shapes] for _ in range(6)]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        return ps, cls([{"params": ps}], lr=3e-3, weight_decay=5e-4, **kw)

    def run(ps, opt, steps):
        for gs in steps:
            opt.zero_grad()
            for p, g in zip(ps, gs):
                if p.grad is None:
                    p.grad = g.clone().cuda()
                else:
                    p.grad.copy_(g.cuda())
            opt.step()

    pt, ot = make(torch.optim.Adam)
    pf, of = make(FusedAdam)
    run(pt, ot, grads[:3]); run(pf, of, grads[:3])
    for a, b in zip(pt, pf):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    sd = of.state_dict()
    assert float(sd["state"][0]["step"]) == 3.0
    for i, p in enumerate(pf):        # the checkpointed moments are the ones the kernel updates (padded offsets)
        np.testing.assert_allclose(sd["state"][i]["exp_avg"].cpu().numpy(), ot.state[pt[i]]["exp_avg"].cpu().numpy(),
                                   rtol=2e-6, atol=1e-8)
    # resume: fresh optimiser on the same parameter values, from FusedAdam's own state and from torch's
    # (deep copies: torch's load_state_dict may alias the tensors of the dict it is given)
    import copy
    sd_t = copy.deepcopy(ot.state_dict())
    for src in (copy.deepcopy(sd), copy.deepcopy(sd_t)):
        pr = [torch.nn.Parameter(p.detach().clone()) for p in pf]
        orr = FusedAdam([{"params": pr}], lr=3e-3, weight_decay=5e-4)
        orr.load_state_dict(src)
        assert orr.step_count == 3
        run(pr, orr, grads[3:])
        pt2 = [torch.nn.Parameter(p.detach().clone()) for p in pt]
        ot2 = torch.optim.Adam([{"params": pt2}], lr=3e-3, weight_decay=5e-4)
        ot2.load_state_dict(copy.deepcopy(sd_t))
        run(pt2, ot2, grads[3:])
        for a, b in zip(pt2, pr):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=3e-6, atol=3e-7)


def test_engine_refreshes_weights_after_load_state_dict():
    """After a first forward the engine computes from compute-dtype weight copies; a later load_state_dict /
    in-place edit of the fp32 masters must be picked up without the caller flagging anything."""
    from oracle import backbone_oracle as bo
    from centroids_reid_amd.baseline import Baseline
    net = Baseline(_cfg(), compute_dtype=torch.float32).cuda().eval()
    x = bo.synthetic_images(2, 64, 32, seed=1).cuda()
    sd_a = bo.make_state_dict("resnet50", 1, seed=11)
    sd_b = bo.make_state_dict("resnet50", 1, seed=12)
    net.base.load_state_dict(sd_a)
    with torch.no_grad():
        _, fa = net(x)
        net.base.load_state_dict(sd_b)
        _, fb = net(x)
        fresh = Baseline(_cfg(), compute_dtype=torch.float32).cuda().eval()
        fresh.base.load_state_dict(sd_b)
        _, fb_ref = fresh(x)
        assert not torch.allclose(fa, fb)
        assert torch.equal(fb, fb_ref)
        net.base.layer4[2].conv3.weight.mul_(0.5)            # in-place edit by "someone else" (EMA, external optimiser)
        _, fc = net(x)
        fresh.base.layer4[2].conv3.weight.mul_(0.5)
        _, fc_ref = fresh(x)
        assert torch.equal(fc, fc_ref) and not torch.equal(fc, fb)


def test_validation_restores_train_mode_and_step_guards_eval_mode():
    from centroids_reid_amd.train_ctl_model import CTLModel
    P, K, C = 4, 4, 12
    model = CTLModel(_cfg(K), num_classes=C, num_query=4, compute_dtype=torch.float32).cuda().train()
    model.configure_optimizers()
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((P * K, 3, 64, 32), generator=gen, device="cuda")
    labels = torch.arange(P, device="cuda").repeat_interleave(K)
    batch = (x, labels, torch.zeros(P * K, dtype=torch.int64), torch.ones(P * K, dtype=torch.bool))
    model.training_step(batch, 0)
    cams = torch.arange(P * K) % 3
    out = model.validation_step((x, labels, cams, torch.arange(P * K)), 0)
    assert not model.backbone.training and not model.bn.training
    with pytest.raises(RuntimeError):
        model.training_step(batch, 1)                       # loud, not a silently frozen backbone
    model.validation_epoch_end([out])
    assert model.backbone.training and model.bn.training
    w0 = model.backbone.base.layer1[0].conv1.weight.detach().clone()
    model.training_step(batch, 1)
    assert not torch.equal(w0, model.backbone.base.layer1[0].conv1.weight.detach())   # the backbone really trains


def test_topk_hits_are_not_truncated_by_max_rank():
    """R1_mAP(max_rank < 50): all_cmc is cut at max_rank but Top-20 / Top-50 use the full match row
    (top_k_retrieval(orig_cmc), utils/eval_reid.py:18-22,84)."""
    from centroids_reid_amd import reid_metric as rm
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(3)
    nq, ng, D = 40, 300, 32
    f = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32))
    pids = rng.integers(0, 60, nq + ng); cams = rng.integers(0, 3, nq + ng)
    full = rm.R1_mAP(num_query=nq, max_rank=50).compute(f.cuda(), pids, cams)
    cut = rm.R1_mAP(num_query=nq, max_rank=10).compute(f.cuda(), pids, cams)
    assert len(cut[0]) == 10
    np.testing.assert_array_equal(cut[0], full[0][:10])
    np.testing.assert_array_equal(cut[2], full[2])
    assert cut[1] == full[1]
    _, _, topk_o, _ = ro.r1_map(f, pids, cams, nq)
    np.testing.assert_allclose(full[2], topk_o, atol=1e-12)
    assert full[2][4] > full[2][2]                           # the case is not degenerate


def test_camset_query_camera_ids_are_validated():
    from centroids_reid_amd import _lib as L, reid_metric as rm
    idx = torch.arange(6, device="cuda").view(2, 3).contiguous()
    with pytest.raises(L.CreidError):
        rm.eval_func(idx, [0, 1], [0, 1, 2], [[64], [0]], [[0], [1], [2]], respect_camids=True)
    with pytest.raises(L.CreidError):
        rm.eval_func(idx, [0, 1], [0, 1, 2], [[-1], [0]], [[0], [1], [2]], respect_camids=True)


def test_compute_chunked_supports_cosine_and_unaligned_feature_width():
    """The reference's `_commpute_batches_double` (utils/reid_metric.py:93-110) works with either SOLVER.DISTANCE_FUNC; here the
    chunked path streams euclidean / fp32 / D % 4 == 0 and walks query chunks for everything else -- same results as compute()."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(8)
    nq, ng = 37, 400
    pids = rng.integers(0, 25, nq + ng); cams = rng.integers(0, 3, nq + ng)
    for D, dist in ((64, "cosine"), (30, "euclidean"), (30, "cosine")):
        f = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32)).cuda()
        ref = rm.R1_mAP(num_query=nq, dist_func=dist).compute(f, pids, cams)
        got = rm.R1_mAP(num_query=nq, dist_func=dist).compute_chunked(f, pids, cams, query_chunk=16)
        np.testing.assert_array_equal(got[0], ref[0])
        assert abs(got[1] - ref[1]) < 1e-12
        np.testing.assert_allclose(got[2], ref[2], atol=1e-12)
        if dist == "euclidean":                         # streamed=True with D % 4 != 0 takes the materialised kernels
            st = rm.R1_mAP(num_query=nq, streamed=True).compute(f, pids, cams)
            assert abs(st[1] - ref[1]) < 1e-12


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_standalone_baseline_returns_base_out_like_the_reference(dtype):
    """modelling/baseline.py:91-96: `Baseline.forward` returns (base_out NCHW, global_feat = GAP(base_out)).  A stand-alone
    Baseline hands the feature map back (a caller that consumes base_out must not get None); inside
    ModelBase / CTLModel, which never read it, the layout pass is switched off."""
    from oracle import backbone_oracle as bo
    from centroids_reid_amd.baseline import Baseline
    net = Baseline(_cfg(), compute_dtype=dtype).cuda()
    x = bo.synthetic_images(2, 64, 32, seed=3).cuda()
    for training in (False, True):
        net.train(training)
        base_out, feat = net(x)
        assert base_out is not None and base_out.dtype == torch.float32 and tuple(base_out.shape) == (2, 2048, 4, 2)
        np.testing.assert_allclose(base_out.mean(dim=(2, 3)).cpu().numpy(), feat.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    from centroids_reid_amd.train_ctl_model import CTLModel
    model = CTLModel(_cfg(), num_classes=8, num_query=0, compute_dtype=dtype).cuda().eval()
    with torch.no_grad():
        base_out, _ = model.backbone(x)
    assert base_out is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,with_res,rb", [(8192, 256, False, 0), (8192, 1024, True, 64), (4096, 2048, True, 7), (1000, 128, False, 3),
                                             (32768, 512, True, 16)])
def test_bn_finalize_apply_one_launch_is_bit_identical_to_two(M, C, with_res, rb, dtype):
    """creid_bn2d_finalize_apply_mask (round 5): training-mode BatchNorm finalize + apply in one launch --
    every apply workgroup sums the partial rows of its own channel strip -- against creid_bn2d_finalize + creid_bn2d_apply_mask:
    output, ReLU bits, mean, invstd, (scale, shift) and the running statistics identical BIT FOR BIT, incl. ragged row blocks."""
    from centroids_reid_amd import _lib as L
    lib = L.lib()
    gen = torch.Generator(device="cuda").manual_seed(M + C)
    rows = (M + 127) // 128
    x = torch.randn((M, C), generator=gen, device="cuda").to(dtype)
    res = torch.randn((M, C), generator=gen, device="cuda").to(dtype) if with_res else None
    pad = rows * 128 - M
    xf = torch.cat([x.float(), torch.zeros((pad, C), device="cuda")]).view(rows, 128, C)
    part = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
    gamma = torch.rand(C, generator=gen, device="cuda") + 0.5
    beta = torch.randn(C, generator=gen, device="cuda") * 0.1
    use_mask = dtype != torch.float32

    def run(fused):
        o = dict(rm=torch.linspace(-0.1, 0.1, C, device="cuda"), rv=torch.linspace(0.5, 1.5, C, device="cuda"),
                 mean=torch.empty(C, device="cuda"), inv=torch.empty(C, device="cuda"), ss=torch.empty((2, C), device="cuda"),
                 y=torch.empty_like(x), mask=torch.zeros(M * C // 8, dtype=torch.uint8, device="cuda"))
        mk = L.ptr(o["mask"]) if use_mask else None
        if fused:
            L.check(lib.creid_bn2d_finalize_apply_mask(L.ptr(part), rows, C, M, L.ptr(o["rm"]), L.ptr(o["rv"]), 0.1, 1e-5, L.ptr(gamma),
                                                       L.ptr(beta), L.ptr(o["mean"]), L.ptr(o["inv"]), L.ptr(o["ss"]), L.ptr(x), L.ptr(res),
                                                       1, L._DT[dtype], L.ptr(o["y"]), mk, rb, L.stream()), "fused")
        else:
            L.check(lib.creid_bn2d_finalize(L.ptr(part), rows, C, M, L.ptr(o["rm"]), L.ptr(o["rv"]), 1, 0.1, 1e-5, L.ptr(gamma),
                                            L.ptr(beta), L.ptr(o["mean"]), L.ptr(o["inv"]), L.ptr(o["ss"]), L.stream()), "fin")
            L.check(lib.creid_bn2d_apply_mask(L.ptr(x), L.ptr(o["ss"]), L.ptr(res), 1, M, C, L._DT[dtype], L.ptr(o["y"]), mk, L.stream()),
                    "apply")
        return o
    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ref = torch.relu((x.float() - x.float().mean(0)) * torch.rsqrt(x.float().var(0, unbiased=False) + 1e-5) * gamma + beta
                     + (res.float() if with_res else 0.0))
    np.testing.assert_allclose(b["y"].float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("arch", ["resnet101", "resnet101_ibn_a"])
def test_deeper_archs_train_through_ctl_model(arch):
    """MODEL.NAME = resnet101 / resnet101_ibn_a (modelling/baseline.py:73-81) through the whole CTLModel step in the bf16 throughput
    mode: two steps, finite losses, every gradient-carrying parameter moves, checkpoint keys as the reference's (631 / 693)."""
    from centroids_reid_amd.bench_train import make_model
    torch.manual_seed(1)
    model = make_model(num_classes=32, dtype=torch.bfloat16, arch=arch)
    assert len(model.state_dict()) == {"resnet101": 631, "resnet101_ibn_a": 693}[arch]
    P, K = 8, 4
    gen = torch.Generator(device="cuda").manual_seed(2)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    for s in range(2):
        x = torch.randn((P * K, 3, 128, 64), generator=gen, device="cuda")
        labels = torch.as_tensor(np.repeat((np.arange(P) * 3 + s) % 32, K).astype(np.int64), device="cuda")
        out = model.training_step((x, labels, torch.zeros(P * K, dtype=torch.int64), torch.ones(P * K, dtype=torch.bool)), s)
        assert np.isfinite(float(out["loss"]))
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters() if n in before)
    assert moved >= 0.98 * len(before), (moved, len(before))
    model.eval()
    with torch.no_grad():
        _, f = model.backbone(torch.randn((4, 3, 128, 64), generator=gen, device="cuda"))
    assert f.shape == (4, 2048) and torch.isfinite(f).all()
