"""Timing ablation of the producer/consumer igemm kernel (run on the GPU box, one process per CREID_IGEMM_ABL value): per
distinct convolution of the B = 64 step, the training forward (BatchNorm statistics epilogue) and the plain data gradient with
the shipped launch plans, 10 back-to-back launches in a graph.  Shapes the plans route to the four-wave DMA kernel do not react."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
os.environ.setdefault("CREID_LIB_PATH", "centroids-reid_amd/lib/libcreid_hip_abl.so")   # python centroids-reid_amd/build.py --ablation
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
from centroids_reid_amd.bench_train import conv_shapes
from bench import time_kernel
B = 64
seen = {}
for cin, cout, k, s, h, w in conv_shapes(B, 256, 128):
    seen[(cin, cout, k, s, h, w)] = seen.get((cin, cout, k, s, h, w), 0) + 1
tf_tot = td_tot = 0.0
for (cin, cout, k, s, h, w), cnt in seen.items():
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    pad = k // 2
    y = ly.conv2d_fwd(x, krsc, s, pad)
    tf = time_kernel(lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), 10) * 1e3
    td = time_kernel(lambda: ly.conv2d_dgrad(y, crsk, (h, w), s, pad), 10) * 1e3
    print(f"{cin:4d}->{cout:4d} k{k} s{s} M={B*y.shape[1]*y.shape[2]:6d} x{cnt}  fwd {tf:6.1f}  dgrad {td:6.1f}")
    tf_tot += tf * cnt; td_tot += td * cnt
print("per-step totals (us): fwd", round(tf_tot), "dgrad", round(td_tot))
