"""Per-shape timing table of the conv kernels over the ResNet50 layer mix (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
from centroids_reid_amd.bench_train import conv_shapes
from bench import time_kernel
B = 64
seen = {}
for cin, cout, k, s, h, w in conv_shapes(B, 256, 128):
    key = (cin, cout, k, s, h, w)
    seen[key] = seen.get(key, 0) + 1
tot = {"f": 0, "d": 0, "w": 0}
print(f"{'shape':34s} cnt  fwd_us  TF/s  GB/s | dgrad_us TF/s | wgrad_us TF/s")
for (cin, cout, k, s, h, w), cnt in seen.items():
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    pad = k // 2
    y = ly.conv2d_fwd(x, krsc, s, pad)
    tf = time_kernel(lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), 10) * 1e3
    td = time_kernel(lambda: ly.conv2d_dgrad(y, crsk, (h, w), s, pad), 10) * 1e3
    dw = torch.zeros((cout, cin, k, k), device="cuda")
    tw = time_kernel(lambda: ly.conv2d_wgrad(x, y, k, s, pad, out=dw), 10) * 1e3
    fl = 2.0 * B * y.shape[1] * y.shape[2] * cout * cin * k * k
    by = (x.numel() + y.numel()) * 2
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {h:3d}x{w:<3d} M={B*y.shape[1]*y.shape[2]:6d} x{cnt}  {tf:7.1f} {fl/tf/1e6:5.0f} {by/tf/1e3:5.0f} | {td:7.1f} {fl/td/1e6:5.0f} | {tw:7.1f} {fl/tw/1e6:5.0f}")
    tot["f"] += tf * cnt; tot["d"] += td * cnt; tot["w"] += tw * cnt
print("per-step totals (us): fwd", round(tot["f"]), "dgrad", round(tot["d"]), "wgrad(+reduce)", round(tot["w"]))
