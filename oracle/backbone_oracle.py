"""CPU oracle for stage A (ResNet50 / ResNet50-IBN-a backbone + GAP + BNNeck).

TEST INFRASTRUCTURE ONLY (see oracle/reid_oracle.py header for the rules).  A clean-room
functional restatement in torch-CPU fp32 of the reference's module graph, driven by a
state_dict whose keys are the reference's (`conv1.weight`, `layer1.0.bn1.running_mean`, ...).
Pinned against the imported reference by tools/gen_golden.py -> tests/golden/backbone_*.npz.

Reference: modelling/backbones/resnet.py:51-133 (Bottleneck, ResNet; NO stem ReLU :97,125),
modelling/backbones/resnet_ibn_a.py:18-141 (IBN split, stem WITH ReLU :129),
modelling/baseline.py:89-96 (GAP).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
# modelling/baseline.py:65-81 (Bottleneck variants of MODEL.NAME) + resnet_ibn_a.py:164-190
ARCH_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3),
               "resnet50_ibn_a": (3, 4, 6, 3), "resnet101_ibn_a": (3, 4, 23, 3),
               "resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}        # BasicBlock networks (modelling/baseline.py:56-65)


def is_basic(arch: str) -> bool:
    return arch in ("resnet18", "resnet34")


def is_ibn(arch: str) -> bool:
    return arch.endswith("_ibn_a")


def layer_strides(last_stride: int = 1):
    return (1, 2, 2, last_stride)


def arch_spec(arch: str = "resnet50", last_stride: int = 1):
    """List of (prefix, inplanes, planes, stride, has_downsample, ibn) for every bottleneck."""
    ibn_arch = is_ibn(arch)
    exp = 1 if is_basic(arch) else 4                    # resnet.py:23,52 BasicBlock.expansion / Bottleneck.expansion
    spec, inpl = [], 64
    for li, (n, pl, st) in enumerate(zip(ARCH_LAYERS[arch], PLANES, layer_strides(last_stride))):
        for b in range(n):
            s = st if b == 0 else 1
            ds = b == 0 and (s != 1 or inpl != pl * exp)
            spec.append((f"layer{li + 1}.{b}", inpl, pl, s, ds, ibn_arch and pl != 512))
            inpl = pl * exp
    return spec


def make_state_dict(arch: str = "resnet50", last_stride: int = 1, seed: int = 1234):
    """Deterministic weights for the whole backbone (reference key names), rebuilt
    identically anywhere from numpy's PCG64 stream (no torch RNG): kaiming-normal convs
    (std sqrt(2/fan_out), as the reference's random_init, resnet.py:156-163), BN gamma
    ~ 1 +- 0.1 (bn3 / downsample gamma ~ 0.5), beta ~ 0.1, running stats perturbed."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = {}

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def conv(name, co, ci, k):
        sd[name + ".weight"] = t(rng.standard_normal((co, ci, k, k)) * math.sqrt(2.0 / (k * k * co)))

    def bn(name, c, gamma=1.0):
        sd[name + ".weight"] = t(gamma * (1.0 + 0.1 * rng.standard_normal(c)))
        sd[name + ".bias"] = t(0.1 * rng.standard_normal(c))
        sd[name + ".running_mean"] = t(0.05 * rng.standard_normal(c))
        sd[name + ".running_var"] = t(1.0 + 0.2 * rng.uniform(-1, 1, c))
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def inorm(name, c):
        sd[name + ".weight"] = t(1.0 + 0.1 * rng.standard_normal(c))
        sd[name + ".bias"] = t(0.1 * rng.standard_normal(c))

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    for pre, inpl, pl, s, ds, ibn in arch_spec(arch, last_stride):
        if is_basic(arch):                              # resnet.py:22-48: conv3x3(stride) - bn - relu - conv3x3 - bn (+ residual) - relu
            conv(pre + ".conv1", pl, inpl, 3)
            bn(pre + ".bn1", pl)
            conv(pre + ".conv2", pl, pl, 3)
            bn(pre + ".bn2", pl, gamma=0.5)
            if ds:
                conv(pre + ".downsample.0", pl, inpl, 1)
                bn(pre + ".downsample.1", pl, gamma=0.5)
            continue
        conv(pre + ".conv1", pl, inpl, 1)
        if ibn:
            inorm(pre + ".bn1.IN", pl // 2)
            bn(pre + ".bn1.BN", pl - pl // 2)
        else:
            bn(pre + ".bn1", pl)
        conv(pre + ".conv2", pl, pl, 3)
        bn(pre + ".bn2", pl)
        conv(pre + ".conv3", pl * 4, pl, 1)
        bn(pre + ".bn3", pl * 4, gamma=0.5)
        if ds:
            conv(pre + ".downsample.0", pl * 4, inpl, 1)
            bn(pre + ".downsample.1", pl * 4, gamma=0.5)
    return sd


def synthetic_images(B, H, W, seed=7):
    """Seeded N(0,1) images [B,3,H,W] fp32 (numpy PCG64; rebuildable on the GPU box)."""
    import numpy as np
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((B, 3, H, W)).astype(np.float32))


def _bn(x, sd, name, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training, momentum, eps)


def _ibn(x, sd, name, training):
    half = sd[name + ".IN.weight"].shape[0]
    xin = x[:, :half]
    xin = xin if xin.dtype == torch.float64 else xin.float()             # (fp64: the yardstick evaluation of the noise-floor tests)
    a = F.instance_norm(xin.contiguous(), None, None, sd[name + ".IN.weight"],
                        sd[name + ".IN.bias"], True, 0.1, 1e-5)        # resnet_ibn_a.py:24,29
    b = _bn(x[:, half:].contiguous(), sd, name + ".BN", training)
    return torch.cat([a, b], 1)


def bottleneck(x, sd, pre, stride, has_ds, ibn, training):
    """resnet.py:67-87 / resnet_ibn_a.py:54-74."""
    out = F.conv2d(x, sd[pre + ".conv1.weight"])
    out = _ibn(out, sd, pre + ".bn1", training) if ibn else _bn(out, sd, pre + ".bn1", training)
    out = F.relu(out)
    out = F.conv2d(out, sd[pre + ".conv2.weight"], stride=stride, padding=1)
    out = F.relu(_bn(out, sd, pre + ".bn2", training))
    out = _bn(F.conv2d(out, sd[pre + ".conv3.weight"]), sd, pre + ".bn3", training)
    res = x
    if has_ds:
        res = _bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], stride=stride), sd,
                  pre + ".downsample.1", training)
    return F.relu(out + res)


def basic_block(x, sd, pre, stride, has_ds, training):
    """resnet.py:22-48 BasicBlock.forward."""
    out = F.relu(_bn(F.conv2d(x, sd[pre + ".conv1.weight"], stride=stride, padding=1), sd, pre + ".bn1", training))
    out = _bn(F.conv2d(out, sd[pre + ".conv2.weight"], padding=1), sd, pre + ".bn2", training)
    res = x
    if has_ds:
        res = _bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], stride=stride), sd, pre + ".downsample.1", training)
    return F.relu(out + res)


def backbone_forward(x, sd, arch="resnet50", last_stride=1, training=False):
    """Returns (base_out [B,2048,h,w], global_feat [B,2048]) -- modelling/baseline.py:91-96 (512 channels for resnet18 / 34).
    In training mode the running stats inside `sd` are updated in place (like nn.BatchNorm2d)."""
    y = _bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1", training)
    if is_ibn(arch):
        y = F.relu(y)                                   # resnet_ibn_a.py:129 (plain ResNets have none)
    y = F.max_pool2d(y, 3, 2, 1)
    for pre, _inpl, _pl, s, ds, ibn in arch_spec(arch, last_stride):
        y = basic_block(y, sd, pre, s, ds, training) if is_basic(arch) else bottleneck(y, sd, pre, s, ds, ibn, training)
    return y, y.mean(dim=(2, 3))


def bnneck_forward(feat, bn_w, bn_b, bn_rm, bn_rv, training):
    """modelling/bases.py:83-84,175-176: BatchNorm1d(2048) on the GAP feature."""
    return F.batch_norm(feat, bn_rm, bn_rv, bn_w, bn_b, training, 0.1, 1e-5)
