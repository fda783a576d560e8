"""Timing ablation of igemm1x1_stream2_kernel (ablation build; one process per CREID_STREAM2_ABL value)."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
os.environ.setdefault("CREID_LIB_PATH", "centroids-reid_amd/lib/libcreid_hip_abl.so")
os.environ["CREID_STREAM2"] = "1"
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
from bench import time_kernel
for B, cin, cout, h, w, mode in ((64, 64, 256, 64, 32, "train"), (128, 64, 256, 64, 32, "eval+res"), (128, 64, 256, 64, 32, "eval"), (128, 128, 512, 32, 16, "eval+res")):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
    krsc, _ = ly.weight_prep(wt, torch.bfloat16)
    ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
    res = torch.randn((B, h, w, cout), device="cuda").to(torch.bfloat16) if mode == "eval+res" else None
    fn = (lambda: ly.conv2d_fwd(x, krsc, 1, 0, with_stats=True)) if mode == "train" else (lambda: ly.conv2d_fwd_affine(x, krsc, 1, 0, ss, res, True))
    t = min(time_kernel(fn, 10) for _ in range(2)) * 1e3
    print(f"{mode:9s} B={B:3d} {cin}->{cout}: {t:6.1f} us", flush=True)
