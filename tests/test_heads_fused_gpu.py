"""GPU: creid_ctl_heads_fused (six multi-role launches, csrc/heads.hip heads_stage*_kernel) against the separate head
launches it replaces (train_ctl_model.py:59-152 between the backbone forward and backward): the arithmetic and the order of
accumulation into the feature gradient are the same, so a whole training step must agree BIT FOR BIT -- losses, logged
statistics, every parameter gradient (backbone included: it sees the pooled-back gradient the last stage writes), BNNeck
running statistics, the lonely-identity counter."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    """Bitwise equality (an overflowing f16 step leaves inf / nan in the gradient buffer: nan != nan under torch.equal)."""
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


def _step(dtype, one_call, is_real, P, K, C, H, W, device_mask, monkeypatch, steps=2):
    from oracle import backbone_oracle as bo
    from centroids_reid_amd import ops
    from centroids_reid_amd.bench_train import make_model
    monkeypatch.setattr(ops, "_DETERMINISTIC", True)        # single-pass classifier GEMMs: no fp32 atomics, bit-reproducible
    torch.manual_seed(0)
    model = make_model(num_classes=C, dtype=dtype, K=K)
    model.heads_one_call = one_call
    model.backbone.base.load_state_dict(bo.make_state_dict("resnet50", 1, seed=11))
    rng = np.random.default_rng(3)
    with torch.no_grad():
        model.center_loss.centers.copy_(torch.from_numpy(rng.standard_normal((C, 2048)).astype(np.float32)) * 0.3)
        model.fc_query.weight.copy_(torch.from_numpy((rng.standard_normal((C, 2048)) * 0.01).astype(np.float32)))
    x = bo.synthetic_images(P * K, H, W, seed=5).cuda()
    labels = torch.from_numpy(np.repeat((np.arange(P) * 3) % C, K).astype(np.int64)).cuda()
    real = torch.as_tensor(is_real)
    if device_mask:
        real = real.cuda()
    out = []
    for s in range(steps):
        o = model.training_step((x, labels, torch.zeros(P * K, dtype=torch.int64), real), s)
        out.append([float(o["loss"])] + [float(v) for v in o["other"].values()] + [float(model.losses_dict[n][-1]) for n in model.losses_names])
    opt, _ = model.optimizers()
    state = {"flat": opt.flat.detach().cpu(), "gflat": opt.gflat.detach().cpu(), "centers": model.center_loss.centers.detach().cpu(),
             "bn_rm": model.bn.running_mean.cpu(), "bn_rv": model.bn.running_var.cpu(), "nbt": int(model.bn.num_batches_tracked),
             "lonely": None if getattr(model, "_lonely_dev", None) is None else int(model._lonely_dev.item())}
    return out, state


@pytest.mark.parametrize("dtype,mask_kind", [(torch.float32, "none"), (torch.bfloat16, "none"), (torch.float16, "none"),
                                             (torch.bfloat16, "device"), (torch.float32, "host"), (torch.float16, "device")])
def test_one_call_heads_equal_separate_launches_bit_for_bit(dtype, mask_kind, monkeypatch):
    P, K, C, H, W = 8, 4, 37, 64, 32
    is_real = np.ones(P * K, dtype=bool)
    if mask_kind != "none":
        is_real[[3, 10, 11, 30]] = False                    # one fake; two fakes in one identity; a fake last slot
    a, sa = _step(dtype, False, is_real, P, K, C, H, W, mask_kind == "device", monkeypatch)
    b, sb = _step(dtype, True, is_real, P, K, C, H, W, mask_kind == "device", monkeypatch)
    assert a == b, (a, b)
    assert sa["nbt"] == sb["nbt"] == 2 and sa["lonely"] == sb["lonely"]
    for k in ("flat", "gflat", "centers", "bn_rm", "bn_rv"):
        assert _same_bits(sa[k], sb[k]), k


def test_one_call_heads_at_the_benchmark_shape(monkeypatch):
    """P = 16 x K = 4, 751 classes, 256 x 128 (BASELINE configs[1]); default split-K classifier GEMMs would differ in the last bits
    between any two runs (fp32 atomics), so this case also runs single-pass."""
    P, K, C, H, W = 16, 4, 751, 256, 128
    is_real = np.ones(P * K, dtype=bool)
    a, sa = _step(torch.bfloat16, False, is_real, P, K, C, H, W, False, monkeypatch, steps=1)
    b, sb = _step(torch.bfloat16, True, is_real, P, K, C, H, W, False, monkeypatch, steps=1)
    assert a == b, (a, b)
    for k in ("flat", "gflat", "centers", "bn_rm", "bn_rv"):
        assert _same_bits(sa[k], sb[k]), k


def test_fused_heads_entry_point_refuses_what_it_does_not_cover():
    """Error behaviour of the C entry point (include/creid.h): argument errors are negative return codes, never a launch --
    K > 16 / D % 8 != 0 -> CREID_E_SHAPE (-4), a workspace that is too small -> CREID_E_WS (-3), missing pointers or a masked
    schedule without its lonely counter -> CREID_E_ARG (-1), the column-sum extension with an fp32 gradient -> CREID_E_SHAPE."""
    import ctypes as C
    from centroids_reid_amd import _lib as L
    lib = L.lib()
    B, P, K, D, Cc, HW = 16, 4, 4, 64, 10, 4
    f = lambda *s: torch.zeros(s, device="cuda")                               # noqa: E731
    feat, centers, bw, bb, rm, rv, W = f(B, D), f(Cc, D), f(D) + 1, f(D), f(D), f(D) + 1, f(Cc, D)
    labels = torch.arange(B, device="cuda") // K
    real = torch.ones(B, dtype=torch.uint8, device="cuda")
    n = 4 * (K + 1) + 2
    wv, stats, g = f(n), f(n + 7), f(B * HW, D)
    nbytes = lib.creid_ctl_heads_workspace_bytes(B, P, K, D, Cc)
    assert nbytes > 0 and lib.creid_ctl_heads_workspace_bytes(0, P, K, D, Cc) == 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def make(**over):
        a = L.CtlHeads()
        a.B, a.P, a.K, a.D, a.num_classes, a.num_centers, a.HW = B, P, K, D, Cc, Cc, HW
        a.g_dtype, a.masked, a.split_logits, a.split_dbnf = 0, 0, 1, 1
        a.margin, a.xent_eps, a.w_query, a.w_center, a.w_xent, a.w_centroid, a.bn_momentum, a.bn_eps = 0.5, 0.1, 1, 5e-4, 1, 1, 0.1, 1e-5
        for name, t in (("feat", feat), ("labels", labels), ("is_real", real), ("centers", centers), ("bn_weight", bw), ("bn_bias", bb),
                        ("bn_running_mean", rm), ("bn_running_var", rv), ("fc_weight", W), ("loss_weights", wv), ("stats", stats),
                        ("g", g), ("workspace", ws)):
            setattr(a, name, t.data_ptr())
        a.workspace_bytes = nbytes
        for k, v in over.items():
            setattr(a, k, v)
        return a

    call = lambda a: lib.creid_ctl_heads_fused(C.byref(a), L.stream())        # noqa: E731
    assert call(make()) == 0                                                     # the covered case runs
    torch.cuda.synchronize()
    assert torch.isfinite(stats).all()
    assert call(make(K=17, P=1)) in (-1, -4)                                     # B != P * K is an argument error first
    assert call(make(workspace_bytes=nbytes // 2)) == -3
    assert call(make(feat=None)) == -1
    assert call(make(masked=1)) == -1                                            # masked schedule without the lonely counter
    assert call(make(g_dtype=7)) == -2
    assert call(make(bn_partial=stats.data_ptr(), bn_x=feat.data_ptr(), bn_mask=real.data_ptr(), bn_mean=bw.data_ptr(),
                     bn_invstd=bw.data_ptr())) == -4                             # column-sum extension: 16-bit gradients only
