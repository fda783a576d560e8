"""conv3x3_c64_kernel against the tile kernels: us per launch at the training and embedding batches (cold operands: the launches
cycle through copies of the input)."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly   # noqa: E402
from bench import time_kernel                  # noqa: E402

for B, H, W in ((64, 64, 32), (128, 64, 32), (56, 80, 80), (256, 80, 80)):
    R = 4 if B < 256 else 2
    xs = [torch.randn((B, H, W, 64), device="cuda").to(torch.bfloat16) for _ in range(R)]
    w = torch.randn((64, 64, 3, 3), device="cuda") / 24
    krsc, _ = ly.weight_prep(w, torch.bfloat16)
    ss = torch.stack([torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1]).contiguous()
    ctr = [0]

    def f_stats():
        ctr[0] += 1
        return ly.conv2d_fwd(xs[ctr[0] % R], krsc, 1, 1, with_stats=True)

    def f_aff():
        ctr[0] += 1
        return ly.conv2d_fwd_affine(xs[ctr[0] % R], krsc, 1, 1, ss, None, True)
    for name, fn in (("stats", f_stats), ("affine", f_aff)):
        res = {}
        for flag in ("0", "1"):
            os.environ["CREID_C64_3X3"] = flag
            res[flag] = min(time_kernel(fn, 12) for _ in range(3)) * 1e3
        if "abl" in os.environ.get("CREID_LIB_PATH", ""):
            os.environ["CREID_C64_3X3"] = "1"
            cells = []
            for a in (0, 1, 2, 4, 8, 3, 7, 15):
                os.environ["CREID_C64_ABL"] = str(a)
                cells.append(f"abl{a}: {min(time_kernel(fn, 12) for _ in range(2)) * 1e3:5.1f}")
            os.environ["CREID_C64_ABL"] = "0"
            print("      " + "  ".join(cells))
        by = 2 * B * H * W * 64 * 2
        print(f"B={B} {H}x{W} {name:6s} tile kernels {res['0']:6.1f} us   halo-tile kernel {res['1']:6.1f} us   ({by / res['1'] / 1e3:.0f} GB/s algorithmic)")
