"""Input side of the path ("device-side augment would follow"): the reference's per-image transforms
(datasets/transforms/build.py:10-33, datasets/transforms/random_erasing.py:11-55) as ONE device pass over a uint8 batch.

    ReidTransforms(cfg).build_transforms(is_train) -> DeviceTransform
    t = DeviceTransform(...)
    params = t.draw(B)                       # host: the reference's random draws, same generators, same order per image
    x = t(images_u8, params)                 # device: flip -> pad -> crop -> ToTensor -> Normalize -> RandomErasing, fp32 NCHW
    x = t(images_u8, params, layout="stem")  # or straight into the stem convolution's padded NHWC4 operand (StemOperand)

The Resize stays on the host (PIL, `DeviceTransform.resize`): it precedes every random draw and works on files of arbitrary
size.  Everything after it is a pure function of (pixels, draws) and runs in `creid_augment_u8` (csrc/augment.hip).
No CPU fallback: a CPU batch raises like every other entry point."""
from __future__ import annotations

import math
import random as _py_random

import numpy as np
import torch

from . import _lib as L


class StemOperand:
    """A batch already in the stem convolution's operand layout (zero-padded NHWC4 [B, H + 8, W + 6, 4], compute dtype);
    `Baseline.forward` / the backbone engine take it in place of the fp32 NCHW tensor and skip their own layout pass."""

    def __init__(self, xpad: torch.Tensor, B: int, H: int, W: int):
        self.xpad, self.B, self.H, self.W = xpad, B, H, W

    @property
    def shape(self):
        return (self.B, 3, self.H, self.W)

    @property
    def device(self):
        return self.xpad.device


class DeviceTransform:
    def __init__(self, size, mean, std, is_train=True, flip_p=0.5, padding=10, re_prob=0.5, sl=0.02, sh=0.4, r1=0.3):
        self.H, self.W = int(size[0]), int(size[1])
        self.mean = [float(v) for v in mean]
        self.std = [float(v) for v in std]
        self.is_train, self.flip_p, self.padding, self.re_prob = bool(is_train), float(flip_p), int(padding), float(re_prob)
        self.sl, self.sh, self.r1 = sl, sh, r1

    # ---- host: Resize (T.Resize(size) on a PIL image = bilinear resize to (W, H))
    def resize(self, pil_image) -> np.ndarray:
        from PIL import Image
        im = pil_image.convert("RGB")
        if im.size != (self.W, self.H):
            im = im.resize((self.W, self.H), Image.BILINEAR)
        return np.asarray(im, dtype=np.uint8)

    # ---- host: the random draws, in the reference pipeline's order for every image
    def draw(self, B: int, rnd=None, generator: torch.Generator | None = None) -> np.ndarray:
        """int32 [B, 8] = {flip, crop_top, crop_left, erase, x1, y1, h, w}.  Flip and crop come from torch's generator
        (torchvision: `torch.rand(1) < p`, `torch.randint(0, h - th + 1, (1,))`, then the column), the erasing rectangle from
        python's `random` (random_erasing.py:33-47) -- pass `rnd` / `generator` to use private streams."""
        rnd = rnd or _py_random
        out = np.zeros((B, 8), np.int32)
        if not self.is_train:
            return out                                                  # the test transform has no draws
        for b in range(B):
            flip = bool(torch.rand(1, generator=generator) < self.flip_p)
            span_h, span_w = 2 * self.padding + 1, 2 * self.padding + 1
            top = int(torch.randint(0, span_h, (1,), generator=generator).item()) if self.padding else 0
            left = int(torch.randint(0, span_w, (1,), generator=generator).item()) if self.padding else 0
            out[b, :3] = (int(flip), top, left)
            out[b, 3:] = self.draw_erasing(rnd)
        return out

    def draw_erasing(self, rnd):
        """random_erasing.py:31-55: (erase, x1 = first row, y1 = first column, h, w)."""
        H, W = self.H, self.W
        if rnd.uniform(0, 1) >= self.re_prob:
            return 0, 0, 0, 0, 0
        for _ in range(100):
            area = H * W
            target_area = rnd.uniform(self.sl, self.sh) * area
            aspect_ratio = rnd.uniform(self.r1, 1 / self.r1)
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < W and h < H:
                return 1, rnd.randint(0, H - h), rnd.randint(0, W - w), h, w
        return 0, 0, 0, 0, 0

    # ---- device
    def __call__(self, images_u8: torch.Tensor, params=None, layout: str = "nchw", dtype=torch.float32):
        """images_u8: uint8 [B, H, W, 3] on the GPU (resized, HWC as PIL yields them).  params: int32 [B, 8] from draw()
        (numpy or device tensor); None = the test transform (and, in training mode, a fresh draw())."""
        L.require_gpu(images_u8)
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3] == 3, "uint8 [B, H, W, 3] expected"
        B, H, W, _ = images_u8.shape
        assert (H, W) == (self.H, self.W), f"images must be resized to {(self.H, self.W)} first (DeviceTransform.resize)"
        images_u8 = images_u8.contiguous()
        if params is None and self.is_train:
            params = self.draw(B)
        pdev = None
        if params is not None and self.is_train:
            pdev = params if isinstance(params, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(params, dtype=np.int32))
            pdev = pdev.to(device=images_u8.device, dtype=torch.int32).contiguous()
            assert tuple(pdev.shape) == (B, 8)
        if layout == "nchw":
            assert dtype == torch.float32, "the reference tensor is fp32"
            out = torch.empty((B, 3, H, W), dtype=torch.float32, device=images_u8.device)
            lay = 0
        elif layout == "stem":
            out = torch.empty((B, H + 8, W + 6, 4), dtype=dtype, device=images_u8.device)
            lay = 1
        else:
            raise ValueError(f"layout {layout!r}: 'nchw' or 'stem'")
        m, s = self.mean, self.std
        L.check(L.lib().creid_augment_u8(L.ptr(images_u8), L.ptr(pdev), B, H, W, self.padding if self.is_train else 0,
                                         m[0], m[1], m[2], s[0], s[1], s[2], m[0], m[1], m[2], lay, L._DT[dtype], L.ptr(out),
                                         L.stream()), "augment_u8")
        return out if lay == 0 else StemOperand(out, B, H, W)


class ReidTransforms:
    """datasets/transforms/build.py:10-33 (same constructor and `build_transforms(is_train)`); the result works on uint8
    batches on the device instead of on one PIL image in a DataLoader worker."""

    def __init__(self, cfg):
        self.cfg = cfg

    def build_transforms(self, is_train=True) -> DeviceTransform:
        i = self.cfg.INPUT
        return DeviceTransform(i.SIZE_TRAIN if is_train else i.SIZE_TEST, i.PIXEL_MEAN, i.PIXEL_STD, is_train=is_train,
                               flip_p=i.PROB, padding=i.PADDING, re_prob=i.RE_PROB)
