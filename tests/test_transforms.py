"""Input transforms (datasets/transforms/build.py:16-31, random_erasing.py:31-55): CPU -- the oracle and the product's host draws
against the reference recording (tests/golden/transforms.npz, made by tools/gen_golden.py `transforms` from the reference's own
RandomErasing class); GPU -- creid_augment_u8 against the oracle and the recording, bit for bit, both output layouts."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import transforms_oracle as to

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz")


def _cases():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        flip, top, left, seed = (int(v) for v in g[f"c{i}_draw"])
        yield i, g[f"c{i}_img"], flip, top, left, seed, float(g[f"c{i}_prob"]), g[f"c{i}_out"], float(g[f"c{i}_next"]), g


def test_oracle_reproduces_the_reference_recording():
    """Same seed -> same erasing rectangle, same number of draws consumed from python's `random` stream, and the transformed
    tensor equals the reference's (torch CPU ToTensor / Normalize arithmetic + the reference's RandomErasing) bit for bit."""
    n_erased = 0
    for i, img, flip, top, left, seed, prob, out, nxt, g in _cases():
        H, W, _ = img.shape
        random.seed(seed)
        er, x1, y1, h, w = to.draw_erasing(random, 3, H, W, probability=prob)
        assert random.random() == nxt, f"case {i}: the oracle consumed a different number of draws"
        t = to.train_transform(img, flip, top, left, er, x1, y1, h, w, int(g["pad"]), g["mean"], g["std"], g["mean"])
        assert t.dtype == np.float32 and np.array_equal(t, out), f"case {i}"
        n_erased += er
    assert 0 < n_erased < int(g["n_cases"])              # both branches of random_erasing.py:33 are in the recording


def test_product_draws_equal_the_oracle_draws():
    from centroids_reid_amd.transforms import DeviceTransform
    for i, img, flip, top, left, seed, prob, out, nxt, g in _cases():
        H, W, _ = img.shape
        t = DeviceTransform((H, W), g["mean"], g["std"], is_train=True, padding=int(g["pad"]), re_prob=prob)
        r1, r2 = random.Random(seed), random.Random(seed)
        assert t.draw_erasing(r1) == to.draw_erasing(r2, 3, H, W, probability=prob)
        assert r1.random() == r2.random()


def test_draw_order_and_ranges():
    """Per image: flip and the two crop offsets from torch's generator, then the erasing draws from python's `random`; the test
    transform draws nothing."""
    from centroids_reid_amd.transforms import DeviceTransform, ReidTransforms
    from centroids_reid_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    tr = ReidTransforms(cfg).build_transforms(is_train=True)
    assert (tr.H, tr.W, tr.padding, tr.flip_p, tr.re_prob) == (256, 128, 10, 0.5, 0.5)
    gen = torch.Generator().manual_seed(3)
    p = tr.draw(64, rnd=random.Random(4), generator=gen)
    assert p.shape == (64, 8) and p.dtype == np.int32
    assert set(np.unique(p[:, 0])) == {0, 1} and p[:, 1:3].min() >= 0 and p[:, 1:3].max() <= 20
    e = p[p[:, 3] == 1]
    assert 0 < len(e) < 64 and (e[:, 4] + e[:, 6] <= 256).all() and (e[:, 5] + e[:, 7] <= 128).all() and (e[:, 6:] > 0).all()
    gen2, r2 = torch.Generator().manual_seed(3), random.Random(4)
    for b in range(3):                                   # the documented order, image by image
        f = bool(torch.rand(1, generator=gen2) < 0.5)
        top = int(torch.randint(0, 21, (1,), generator=gen2)); left = int(torch.randint(0, 21, (1,), generator=gen2))
        assert (int(f), top, left) == tuple(p[b, :3])
        assert tr.draw_erasing(r2) == tuple(p[b, 3:])
    te = ReidTransforms(cfg).build_transforms(is_train=False)
    st = random.getstate()
    assert not te.draw(5).any() and random.getstate() == st
    with pytest.raises(L_error()):
        tr(torch.zeros((2, 256, 128, 3), dtype=torch.uint8))          # no CPU fallback


def L_error():
    from centroids_reid_amd import _lib as L
    return L.CreidError


@pytest.mark.gpu
def test_device_transform_equals_reference_recording():
    from centroids_reid_amd.transforms import DeviceTransform
    for i, img, flip, top, left, seed, prob, out, nxt, g in _cases():
        H, W, _ = img.shape
        t = DeviceTransform((H, W), g["mean"], g["std"], is_train=True, padding=int(g["pad"]), re_prob=prob)
        er = t.draw_erasing(random.Random(seed))
        params = np.array([[flip, top, left, *er]], np.int32)
        y = t(torch.from_numpy(img[None]).cuda(), params)
        assert y.dtype == torch.float32 and np.array_equal(y[0].cpu().numpy(), out), f"case {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,B", [(256, 128, 9), (320, 320, 3), (33, 17, 5), (8, 8, 2)])
def test_device_transform_equals_oracle_all_layouts(H, W, B):
    """Random draws incl. the extreme crop offsets (0 and 2 * pad: the black border on either side), erasing rectangles touching
    the borders, no erasing; fp32 NCHW bit-exact against the oracle; the stem-operand layout equals creid_image_to_nhwc4_pad of
    the NCHW result in fp32 and bf16; the test transform (no params) = ToTensor + Normalize."""
    import ctypes as C
    from centroids_reid_amd import _lib as L
    from centroids_reid_amd.transforms import DeviceTransform, StemOperand
    rng = np.random.default_rng(H * 7 + W)
    mean, std, pad = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], 10
    imgs = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    imgs[0] = 0; imgs[-1] = 255
    t = DeviceTransform((H, W), mean, std, is_train=True, padding=pad)
    params = t.draw(B, rnd=random.Random(1), generator=torch.Generator().manual_seed(2))
    params[0, :3] = (1, 0, 0); params[1, :3] = (0, 2 * pad, 2 * pad)
    params[0, 3:] = (1, 0, 0, H - 1, W - 1)                          # the largest block the reference can draw (h < H, w < W)
    params[1, 3:] = (1, H - 2, W - 3, 2, 3)                          # touching the bottom-right corner
    if B > 2:
        params[2, 3:] = (0, 5, 5, 4, 4)                              # flag off: the rectangle is ignored
    dev = torch.from_numpy(imgs).cuda()
    y = t(dev, params).cpu().numpy()
    for b in range(B):
        ref = to.train_transform(imgs[b], *[int(v) for v in params[b]], pad, mean, std, mean)
        assert np.array_equal(y[b], ref), b
    lib = L.lib()
    for dt in (torch.float32, torch.bfloat16):
        s = t(dev, params, layout="stem", dtype=dt)
        assert isinstance(s, StemOperand) and tuple(s.xpad.shape) == (B, H + 8, W + 6, 4) and s.xpad.dtype == dt
        want = torch.empty_like(s.xpad)
        L.check(lib.creid_image_to_nhwc4_pad(L.ptr(torch.from_numpy(y).cuda()), B, H, W, L._DT[dt], L.ptr(want), L.stream()), "pad")
        torch.cuda.synchronize()
        assert torch.equal(s.xpad, want), dt
    te = DeviceTransform((H, W), mean, std, is_train=False)
    yt = te(dev).cpu().numpy()
    for b in range(B):
        assert np.array_equal(yt[b], to.test_transform(imgs[b], mean, std)), b


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backbone_takes_the_stem_operand(dtype):
    """Baseline.forward on a transforms.StemOperand == Baseline.forward on the fp32 NCHW batch of the same transform: eval mode
    (folded forward) and training mode (forward + one backward: same embedding, same stem weight gradient), bit for bit."""
    from oracle import backbone_oracle as bo
    from centroids_reid_amd.bench_train import make_model
    from centroids_reid_amd.transforms import DeviceTransform
    B, H, W = 4, 64, 32
    model = make_model(num_classes=10, dtype=dtype, K=2)
    model.backbone.base.load_state_dict(bo.make_state_dict("resnet50", 1, seed=3))
    model.backbone.base.cuda()
    imgs = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    t = DeviceTransform((H, W), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], is_train=True, padding=10)
    params = t.draw(B, rnd=random.Random(0), generator=torch.Generator().manual_seed(0))
    x_nchw, x_stem = t(imgs, params), t(imgs, params, layout="stem", dtype=dtype)
    model.eval()
    with torch.no_grad():
        _, f0 = model.backbone(x_nchw)
        _, f1 = model.backbone(x_stem)
    assert torch.equal(f0, f1)
    model.train()
    grads = []
    for x in (x_nchw, x_stem):
        model.zero_grad(set_to_none=True)
        _, f = model.backbone(x)
        f.square().sum().backward()
        torch.cuda.synchronize()
        grads.append((f.detach().clone(), model.backbone.base.conv1.weight.grad.detach().clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.gpu
def test_run_inference_takes_uint8_batches():
    """inference.run_inference on a loader of uint8 [B, H, W, 3] batches (test transform on the device, straight into the stem
    operand) == on the reference-style loader of fp32 NCHW tensors made by the oracle's ToTensor + Normalize: same embeddings, bit
    for bit, same paths; a uint8 batch without a transform is refused."""
    from oracle import backbone_oracle as bo
    from centroids_reid_amd import inference as inf
    from centroids_reid_amd.bench_train import make_model
    from centroids_reid_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.INPUT.SIZE_TEST = [64, 32]
    model = make_model(num_classes=10, dtype=torch.float32, K=2)
    model.backbone.base.load_state_dict(bo.make_state_dict("resnet50", 1, seed=4))
    model.backbone.base.cuda()
    rng = np.random.default_rng(1)
    u8 = [rng.integers(0, 256, (3, 64, 32, 3), dtype=np.uint8) for _ in range(2)]
    paths = [[f"img_{b}_{i}.jpg" for i in range(3)] for b in range(2)]
    mean, std = cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD
    ref_loader = [(torch.from_numpy(np.stack([to.test_transform(im, mean, std) for im in b])), "", p) for b, p in zip(u8, paths)]
    u8_loader = [(torch.from_numpy(b), "", p) for b, p in zip(u8, paths)]
    e0, p0 = inf.run_inference(model, ref_loader, cfg)
    e1, p1 = inf.run_inference(model, u8_loader, cfg)
    assert np.array_equal(e0, e1) and list(p0) == list(p1) and e0.shape == (6, 2048)
    with pytest.raises(ValueError):
        inf._inference(model, u8_loader[0])
