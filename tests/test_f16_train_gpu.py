"""f16 mixed-precision TRAINING -- the reference's own mixed precision (utils/misc.py:111 `precision=16`: native AMP with a
GradScaler under pytorch-lightning 1.1.4) -- on the HIP step: f16 MFMA inputs / activations / gradient tensors, fp32 master
weights, accumulators, heads and optimiser, dynamic loss scale with all of its state on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dtype, num_classes=64, seed=0):
    from centroids_reid_amd.bench_train import make_model
    torch.manual_seed(seed)
    return make_model(num_classes=num_classes, dtype=dtype)


def _batches(P, K, H, W, steps, n_id=64, noise=0.6):
    gen = torch.Generator(device="cuda").manual_seed(1)
    base = torch.randn((n_id, 3, H // 16, W // 16), generator=gen, device="cuda")
    base = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    g2 = torch.Generator(device="cuda").manual_seed(2)
    out = []
    for s in range(steps):
        ids = (np.arange(P) * 5 + s * P) % n_id
        x = base[torch.as_tensor(ids, device="cuda")].repeat_interleave(K, 0) + noise * torch.randn((P * K, 3, H, W), generator=g2, device="cuda")
        out.append((x, torch.as_tensor(np.repeat(ids, K).astype(np.int64), device="cuda"), torch.zeros(P * K, dtype=torch.int64),
                    torch.ones(P * K, dtype=torch.bool)))
    return out


def test_amp_kernels_scale_unscale_skip_update():
    """The device-side GradScaler: scale on the way in, in-place unscale + non-finite check, the skipped Adam / SGD step (step
    counter included), backoff on overflow, growth after `interval` clean steps -- every decision taken by the kernels."""
    from centroids_reid_amd.solver import LossScaler, FusedAdam, CenterSGD
    sc = LossScaler("cuda", init_scale=1024.0, growth_interval=3)
    x = torch.randn(1000, device="cuda")
    np.testing.assert_array_equal(sc.scale_(x).cpu().numpy(), (x * 1024.0).cpu().numpy())
    p = torch.nn.Parameter(torch.randn(4096, device="cuda"))
    q = torch.nn.Parameter(torch.randn(100, device="cuda"))          # an unscaled (head) parameter behind the scaled prefix
    c = torch.nn.Parameter(torch.randn(64, device="cuda"))
    opt = FusedAdam([{"params": [p, q], "names": ["backbone.w", "fc.w"]}], lr=1e-2)
    opt.attach_scaler(sc)
    assert opt.n_scaled == 4096
    optc = CenterSGD([{"params": [c], "names": ["center"]}], lr=0.5)
    optc.scaler = sc
    g_true = torch.randn(4096, device="cuda")
    ref = torch.optim.Adam([torch.nn.Parameter(torch.cat([p.detach(), q.detach()]).clone())], lr=1e-2)
    gq = torch.randn(100, device="cuda")
    cg = torch.randn(64, device="cuda")
    c0 = c.detach().clone()
    for step, poison in enumerate([False, True, False, False, False]):
        p.grad.copy_(g_true * sc.get_scale()); q.grad.copy_(gq); c.grad = cg.clone()
        if poison:
            p.grad[17] = float("inf")
        before = (p.detach().clone(), q.detach().clone(), c.detach().clone(), opt.step_count, sc.get_scale())
        opt.step(); optc.step(); sc.update()
        if poison:
            assert torch.equal(p.detach(), before[0]) and torch.equal(q.detach(), before[1]) and torch.equal(c.detach(), before[2])
            assert opt.step_count == before[3] and sc.get_scale() == before[4] * 0.5
        else:
            rp = ref.param_groups[0]["params"][0]
            rp.grad = torch.cat([g_true, gq])
            ref.step()
            np.testing.assert_allclose(torch.cat([p.detach(), q.detach()]).cpu().numpy(), rp.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(p.grad.cpu().numpy(), g_true.cpu().numpy(), rtol=1e-6, atol=0)     # left unscaled, like unscale_()
            assert opt.step_count == before[3] + 1
    # 1024 -> 512 (overflow at step 1) -> three clean steps -> 1024
    assert sc.get_scale() == 1024.0 and int(sc.flags[0]) == 0
    assert not torch.equal(c.detach(), c0)


def test_f16_ctl_training_trajectory_tracks_fp32():
    """`CTLModel(cfg, compute_dtype=torch.float16)` trains -- 20 full steps (four losses, backward, Adam +
    center SGD, loss scale 65536 as GradScaler starts) from the same seed on the same clustered batches as the exact-f32
    parity mode: loss trajectory within 2 % at every step, no step skipped, scale untouched."""
    P, K, H, W, steps = 16, 4, 256, 128, 20
    batches = _batches(P, K, H, W, steps)
    curves = {}
    for dt in (torch.float32, torch.float16):
        model = _model(dt)
        assert (getattr(model, "loss_scaler", None) is not None) == (dt == torch.float16)
        losses = [model.training_step(b, s)["loss"].detach().float().reshape(()) for s, b in enumerate(batches)]
        curves[dt] = torch.stack(losses).cpu().numpy().astype(np.float64)
        if dt == torch.float16:
            opt, _ = model.optimizers()
            assert opt.step_count == steps and model.loss_scaler.get_scale() == 65536.0
            assert all(torch.isfinite(p).all() for p in model.parameters())
    rel = np.abs(curves[torch.float16] - curves[torch.float32]) / np.abs(curves[torch.float32])
    assert rel.max() < 0.02, (rel.max(), curves)
    assert curves[torch.float16][-1] < curves[torch.float16][0] - 0.5            # and it learns


def test_f16_step_in_a_captured_graph_survives_an_overflow(monkeypatch):
    """The f16 step captured ONCE into a hipGraph: replays equal eager steps bit for bit, and a replay whose gradients overflow
    (the scale forced to 2^24 on the device) skips its update, halves the scale and the following replays train on -- no host
    involvement between replays."""
    from centroids_reid_amd import ops
    from centroids_reid_amd.bench_train import CAPTURE_MODE
    monkeypatch.setattr(ops, "_DETERMINISTIC", True)          # single-pass classifier GEMMs: bit-reproducible steps
    P, K, H, W = 8, 4, 128, 64
    batches = _batches(P, K, H, W, 6)
    def fresh():
        m = _model(torch.float16)
        # GradScaler's start value (65536) is meant to overflow and back off during the first steps of a run; a settled scale keeps
        # this test about the mechanism: no step below is skipped unless the test forces it
        m.loss_scaler.state.copy_(torch.tensor([1024.0, 1.0 / 1024.0]))
        return m
    eager = fresh()
    for s in range(5):
        eager.training_step(batches[s], s)
    assert eager.optimizers()[0].step_count == 5 and eager.loss_scaler.get_scale() == 1024.0
    model = fresh()
    sx, sl = batches[0][0].clone(), batches[0][1].clone()
    static = (sx, sl, batches[0][2], batches[0][3])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in range(2):
            sx.copy_(batches[s][0]); sl.copy_(batches[s][1])
            model.training_step(static, s)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
        out = model.training_step(static, 0)
    # (the capture itself enqueued nothing: steps 2..4 are replays)
    for s in range(2, 5):
        sx.copy_(batches[s][0]); sl.copy_(batches[s][1])
        graph.replay()
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(eager.named_parameters(), model.named_parameters()):
        assert torch.equal(a, b), n
    # force an overflow: with a scale of 2^30 the very first f16 gradient tensor saturates
    sc = model.loss_scaler
    sc.state.copy_(torch.tensor([2.0 ** 30, 2.0 ** -30]))
    opt, _ = model.optimizers()
    before = opt.flat.clone(); n0 = opt.step_count
    sx.copy_(batches[5][0]); sl.copy_(batches[5][1])
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(opt.flat, before) and opt.step_count == n0 and sc.get_scale() == 16777216.0    # halved, then clamped to 2^24
    sc.state.copy_(torch.tensor([1024.0, 1.0 / 1024.0]))
    graph.replay(); torch.cuda.synchronize()
    assert not torch.equal(opt.flat, before) and opt.step_count == n0 + 1 and np.isfinite(float(out["loss"]))


def test_f16_ibn_a_step_and_checkpoint_scaler_state():
    """ResNet50-IBN-a (the Street2Shop / DukeMTMC canonical configs turn precision=16 on for it) trains in f16 too: three full steps
    are finite and applied; the checkpoint carries the loss-scale state under pytorch-lightning's key and a reloaded scaler resumes
    from it."""
    from centroids_reid_amd.bench_train import make_model
    from centroids_reid_amd.solver import LossScaler
    torch.manual_seed(0)
    model = make_model(num_classes=64, dtype=torch.float16, arch="resnet50_ibn_a")
    model.loss_scaler.state.copy_(torch.tensor([1024.0, 1.0 / 1024.0]))
    for s, b in enumerate(_batches(8, 4, 128, 64, 3)):
        out = model.training_step(b, s)
        assert np.isfinite(float(out["loss"]))
    opt, _ = model.optimizers()
    assert opt.step_count == 3 and all(torch.isfinite(p).all() for p in model.parameters())
    ck = model.checkpoint_dict()
    st = ck["native_amp_scaling_state"]
    assert st["scale"] == 1024.0 and st["growth_interval"] == 2000 and st["_growth_tracker"] == 3
    sc = LossScaler("cuda")
    sc.load_state_dict(st)
    assert sc.get_scale() == 1024.0 and int(sc.flags[1]) == 3 and int(sc.flags[0]) == 0
    # resume (what pytorch-lightning's restore_training_state does): a fresh module + load_training_state continues from the same
    # Adam step, moments, loss scale and growth tracker (the scaler state used to be written but never read back)
    model2 = make_model(num_classes=64, dtype=torch.float16, arch="resnet50_ibn_a")
    model2.load_state_dict(ck["state_dict"])
    model2.load_training_state(ck)
    opt2, _ = model2.optimizers()
    assert opt2.step_count == 3 and model2.loss_scaler.get_scale() == 1024.0 and int(model2.loss_scaler.flags[1]) == 3
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)


def test_head_gradient_overflow_skips_the_step_and_counts_it():
    """GradScaler.step inspects EVERY parameter of the optimiser: a non-finite gradient confined to the heads (here: the
    classifier) must skip Adam and the center SGD like a backbone overflow does, and the skipped step is visible on the host."""
    from centroids_reid_amd.bench_train import make_model
    torch.manual_seed(0)
    model = make_model(num_classes=64, dtype=torch.float16)
    model.loss_scaler.state.copy_(torch.tensor([1024.0, 1.0 / 1024.0]))
    b = _batches(8, 4, 64, 32, 1)[0]
    model.training_step(b, 0)
    opt, _ = model.optimizers()
    assert opt.step_count == 1 and model.loss_scaler.skipped_steps == 0
    model.forward_backward(b, 1)
    model.fc_query.weight.grad[3, 5] = float("inf")
    flat0, centers0 = opt.flat.clone(), model.center_loss.centers.detach().clone()
    model.apply_optimizers()
    assert opt.step_count == 1 and torch.equal(opt.flat, flat0) and torch.equal(model.center_loss.centers.detach(), centers0)
    assert model.loss_scaler.skipped_steps == 1 and model.loss_scaler.get_scale() == 512.0
