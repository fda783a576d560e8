"""Would the persistent all-waves-multiply kernel (conv_pipe.hip) pay for the DATA GRADIENTS of layer3 / layer4 at the training
batch, if it had the fused epilogues (BatchNorm column sums, masked residual add) the tile kernels carry there?  Upper bound without
building them: the PLAIN data gradient (no fused pass) through the tile kernels against every persistent variant, isolated, warm,
10 launches per graph.  The fused passes cost the same bytes in either kernel, so a plain-vs-plain win is the most the port could give.
    python tools/debug/pp_dgrad_probe.py [batch]"""
import os
os.environ.setdefault("CREID_DEBUG_KNOBS", "1")
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly   # noqa: E402
from bench import time_kernel                  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def vword(bm, bn, kph, mode):
    return (bm // 128) | ((bn // 128) << 2) | (kph << 4) | (mode << 8)


VAR = [(bm, bn, kph, mode) for (bm, bn) in ((256, 256), (128, 256), (256, 128), (128, 128)) for (kph, mode) in ((1, 0), (2, 0), (1, 2))]
tot_t = tot_p = 0.0
print(f"B={B}: plain data gradient, us: tile kernels (rule / plan) | best persistent variant | ratio")
for name, cin, cout, k, s, h, w, cnt in (("L3 c1", 1024, 256, 1, 1, 16, 8, 5), ("L3 c2", 256, 256, 3, 1, 16, 8, 5), ("L3 c3", 256, 1024, 1, 1, 16, 8, 6),
                                         ("L4 c1", 2048, 512, 1, 1, 16, 8, 2), ("L4 c2", 512, 512, 3, 1, 16, 8, 3), ("L4 c3", 512, 2048, 1, 1, 16, 8, 3),
                                         ("L4 ds", 1024, 2048, 1, 1, 16, 8, 1), ("L4.0 c1", 1024, 512, 1, 1, 16, 8, 1)):
    pad = k // 2
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    dy = torch.randn((B, h, w, cout), device="cuda").to(torch.bfloat16)
    os.environ["CREID_IGEMM_PP"] = "0"
    t_tile = min(time_kernel(lambda: ly.conv2d_dgrad(dy, crsk, (h, w), s, pad), 10) for _ in range(2)) * 1e3
    best = (1e9, None)
    for v in VAR:
        if cin % v[1]:
            continue
        os.environ["CREID_IGEMM_PP"] = hex(0x1000 | vword(*v))
        t = min(time_kernel(lambda: ly.conv2d_dgrad(dy, crsk, (h, w), s, pad), 10) for _ in range(2)) * 1e3
        if t < best[0]:
            best = (t, v)
    tot_t += t_tile * cnt; tot_p += min(best[0], t_tile) * cnt
    print(f"{name:8s} {cout:4d}->{cin:4d} k{k} x{cnt}: {t_tile:6.1f} | {best[0]:6.1f} {best[1]} | {best[0] / t_tile:4.2f}", flush=True)
print(f"per step: tile kernels {tot_t:.0f} us, best of both {tot_p:.0f} us")
