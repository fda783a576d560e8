import sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import layers as ly, _lib as L
lib = L.lib()
B = 64
for cin, cout, k, h, w in ((512, 512, 3, 16, 8), (512, 2048, 1, 16, 8), (1024, 2048, 1, 16, 8), (2048, 512, 1, 16, 8), (256, 256, 3, 16, 8), (1024, 256, 1, 16, 8)):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    y = ly.conv2d_fwd(x, krsc, 1, k // 2)
    M, K = B * h * w, k * k * cin
    for tag, fn, key in (("fwd", lambda: ly.conv2d_fwd(x, krsc, 1, k // 2, with_stats=True), (M, cout, K, 0)),
                         ("dgrad", lambda: ly.conv2d_dgrad(y, crsk, (h, w), 1, k // 2), (M, cin, k * k * cout, 1))):
        res = []
        for bn, st in ((64, 2), (64, 3), (64, 4), (128, 2), (128, 3), (128, 4)):
            if key[1] % bn:
                continue
            lib.creid_tune_clear()
            lib.creid_tune_set(1, *key, bn, st, 0)
            res.append(f"({bn},{st}) {min(time_kernel(fn, 10) for _ in range(2)) * 1e3:5.1f}")
        lib.creid_tune_clear()
        print(f"{tag:5s} {cin}->{cout} k{k}: " + "  ".join(res), flush=True)
