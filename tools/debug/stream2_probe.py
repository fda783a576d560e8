"""Per-shape timing of the second persistent 1x1 kernel against the shipped plans (run on the GPU box): the K = 64 / 128 stride-1
1x1 layers of ResNet50 256x128 -- training forward with the statistics epilogue at B = 64, folded eval-mode epilogue (conv3: +
residual + ReLU) at B = 128."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
from bench import time_kernel

SHAPES = [(64, 64, 64, 32, "c1"), (64, 256, 64, 32, "c3"), (128, 512, 32, 16, "c3"), (256, 64, 64, 32, "c1"), (256, 128, 64, 32, "c1"),
          (256, 1024, 16, 8, "c3")]


def t_us(fn):
    return min(time_kernel(fn, 10) for _ in range(2)) * 1e3


for B, mode in ((64, "train"), (128, "eval")):
    for cin, cout, h, w, role in SHAPES:
        x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
        wt = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
        krsc, _ = ly.weight_prep(wt, torch.bfloat16)
        ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
        res = torch.randn((B, h, w, cout), device="cuda").to(torch.bfloat16) if role == "c3" else None
        if mode == "train":
            fn = lambda: ly.conv2d_fwd(x, krsc, 1, 0, with_stats=True)
            byt = (x.numel() + B * h * w * cout) * 2
        else:
            fn = lambda: ly.conv2d_fwd_affine(x, krsc, 1, 0, ss, res, True)
            byt = (x.numel() + B * h * w * cout * (2 if res is not None else 1)) * 2
        out = []
        for env in ({"CREID_STREAM2": "0"}, {"CREID_STREAM2": "1"}, {"CREID_STREAM2": "1", "CREID_STREAM2_BN": "128"},
                    {"CREID_STREAM2": "1", "CREID_STREAM2_BN": "64"}):
            os.environ.pop("CREID_STREAM2_BN", None)
            os.environ.update(env)
            t = t_us(fn)
            out.append(f"{t:6.1f} us {byt / t / 1e3:5.0f} GB/s")
        print(f"{mode:5s} B={B:3d} {cin:3d}->{cout:3d} {h}x{w}: shipped {out[0]} | stream2 {out[1]} | bn128 {out[2]} | bn64 {out[3]}", flush=True)
