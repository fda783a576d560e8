"""TEST INFRASTRUCTURE (CPU oracle, never imported by the product): numpy restatement of the reference's per-image input
transforms after the Resize, datasets/transforms/build.py:16-31 and datasets/transforms/random_erasing.py:31-55.

Pinning: RandomErasing (its draws from python's `random` and its write into the normalised tensor) is checked against the
reference's own class, imported from /root/reference by tools/gen_golden.py `transforms` -> tests/golden/transforms.npz.
RandomHorizontalFlip / Pad / RandomCrop / ToTensor / Normalize are torchvision classes and torchvision is absent from this image:
PARITY UNPINNED for those five -- they are restated from torchvision's published semantics (flip = reverse the width axis;
Pad(p) = constant 0 border on the uint8 image; crop = slice; ToTensor = HWC uint8 -> CHW float32 / 255; Normalize =
(t - mean) / std in fp32), and the golden file pins the fp32 arithmetic against torch's own CPU ops on the same bytes."""
import math

import numpy as np


def draw_erasing(rnd, C, H, W, probability=0.5, sl=0.02, sh=0.4, r1=0.3):
    """The draws of RandomErasing.__call__ (random_erasing.py:31-55) from `rnd` (the `random` module or a random.Random):
    returns (erase, x1, y1, h, w) with x1 the first ROW and y1 the first COLUMN, like the reference's names."""
    if rnd.uniform(0, 1) >= probability:                               # :33
        return 0, 0, 0, 0, 0
    for _attempt in range(100):                                        # :36
        area = H * W                                                   # :37
        target_area = rnd.uniform(sl, sh) * area                       # :39
        aspect_ratio = rnd.uniform(r1, 1 / r1)                         # :40
        h = int(round(math.sqrt(target_area * aspect_ratio)))          # :42
        w = int(round(math.sqrt(target_area / aspect_ratio)))          # :43
        if w < W and h < H:                                            # :45
            x1 = rnd.randint(0, H - h)                                 # :46
            y1 = rnd.randint(0, W - w)                                 # :47
            return 1, x1, y1, h, w
    return 0, 0, 0, 0, 0                                               # :55


def to_tensor_normalize(img_u8_hwc, mean, std):
    """T.ToTensor() + T.Normalize(mean, std) (build.py:16,23-24): float32 CHW."""
    t = np.ascontiguousarray(img_u8_hwc.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def train_transform(img_u8_hwc, flip, top, left, erase, x1, y1, h, w, pad, mean, std, erase_value):
    """build.py:18-26 on one already-resized image [H, W, 3] uint8 with explicit draws; returns float32 [3, H, W]."""
    H, W, _ = img_u8_hwc.shape
    im = img_u8_hwc[:, ::-1] if flip else img_u8_hwc                   # T.RandomHorizontalFlip (:19)
    im = np.pad(im, ((pad, pad), (pad, pad), (0, 0)))                  # T.Pad(padding), fill 0 (:20)
    im = im[top:top + H, left:left + W]                                # T.RandomCrop(size) at (top, left) (:21)
    t = to_tensor_normalize(im, mean, std)                             # :22-23
    if erase:                                                          # RandomErasing (:24; random_erasing.py:48-53)
        for c in range(3):
            t[c, x1:x1 + h, y1:y1 + w] = np.float32(erase_value[c])
    return t


def test_transform(img_u8_hwc, mean, std):
    """build.py:27-31 after the Resize."""
    return to_tensor_normalize(img_u8_hwc, mean, std)


def stem_operand(t_chw, dtype_bits=32):
    """The stem convolution's operand the product can emit directly: zero-padded NHWC4 [H + 8, W + 6, 4], image at (3, 3)."""
    C, H, W = t_chw.shape
    out = np.zeros((H + 8, W + 6, 4), np.float32)
    out[3:3 + H, 3:3 + W, :3] = t_chw.transpose(1, 2, 0)
    return out
