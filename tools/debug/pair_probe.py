"""us per call of the one-launch layer1 block boundary (conv_pair.hip) against the two launches it replaces, cold operands."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centroids_reid_amd import _lib as L
from centroids_reid_amd import layers as ly
lib = L.lib()
for B, H, W in ((128, 64, 32), (256, 80, 80)):
    R = 3 if B == 128 else 2
    dt = torch.bfloat16
    a2 = [torch.randn((B, H, W, 64), device="cuda").to(dt) for _ in range(R)]
    res = [torch.randn((B, H, W, 256), device="cuda").to(dt) for _ in range(R)]
    k3, _ = ly.weight_prep(torch.randn((256, 64, 1, 1), device="cuda") / 8, dt)
    k1, _ = ly.weight_prep(torch.randn((64, 256, 1, 1), device="cuda") / 16, dt)
    ss3 = torch.rand((2, 256), device="cuda") + 0.5
    ss1 = torch.rand((2, 64), device="cuda") + 0.5
    out3 = torch.empty((B, H, W, 256), device="cuda", dtype=dt)
    out1 = torch.empty((B, H, W, 64), device="cuda", dtype=dt)
    M = B * H * W
    ctr = [0]

    def two():
        ctr[0] += 1
        y = ly.conv2d_fwd_affine(a2[ctr[0] % R], k3, 1, 0, ss3, res[ctr[0] % R], True)
        return ly.conv2d_fwd_affine(y, k1, 1, 0, ss1, None, True)

    def one():
        ctr[0] += 1
        L.check(lib.creid_bottleneck_c3_c1_fwd_affine(M, 64, 256, 64, L.ptr(a2[ctr[0] % R]), L.ptr(k3), L.ptr(ss3), L.ptr(res[ctr[0] % R]),
                                                      L.ptr(out3), L.ptr(k1), L.ptr(ss1), L.ptr(out1), L.dtype_code(out1), L.stream()), "pair")

    def timeit(fn, n=12):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t2, t1 = min(timeit(two) for _ in range(2)), min(timeit(one) for _ in range(2))
    by = M * (64 + 256 + 256 + 64) * 2
    print(f"B={B} {H}x{W}: two launches {t2:.1f} us, one launch {t1:.1f} us ({by / t1 / 1e3:.0f} GB/s of its {by / 1e6:.0f} MB)", flush=True)
    for wgs in (128, 512):
        os.environ["CREID_STREAM1X1_WGS"] = str(wgs)
        print(f"   wgs {wgs}: {timeit(one):.1f} us", flush=True)
    os.environ.pop("CREID_STREAM1X1_WGS")
