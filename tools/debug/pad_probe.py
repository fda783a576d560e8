"""us per call of the image layout pass (fp32 NCHW -> padded NHWC4 in the compute dtype) at the bench shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centroids_reid_amd import _lib as L
lib = L.lib()
for B, H, W in ((64, 256, 128), (128, 256, 128), (256, 320, 320)):
    x = torch.randn((B, 3, H, W), device="cuda")
    y = torch.empty((B, H + 8, W + 6, 4), device="cuda", dtype=torch.bfloat16)
    dt, st = L.dtype_code(y), L.stream()
    f = lambda: L.check(lib.creid_image_to_nhwc4_pad(L.ptr(x), B, H, W, dt, L.ptr(y), st), "pad")
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B} {H}x{W}: {us:.1f} us  ({(x.numel() * 4 + y.numel() * 2) / us / 1e3:.0f} GB/s)")
