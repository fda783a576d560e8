"""Timing ablation of the weight-gradient partial-tile kernel (run on the GPU box, one process per CREID_WGRAD_ABL value):
per distinct convolution of the B = 64 step, the partial-tile launch alone (no split reduce), 10 back-to-back launches in a graph."""
import ctypes as C
import os
import sys
import torch
os.environ.setdefault("CREID_LIB_PATH", "centroids-reid_amd/lib/libcreid_hip_abl.so")   # python centroids-reid_amd/build.py --ablation
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L
from centroids_reid_amd.bench_train import conv_shapes
from bench import time_kernel
B = 64
seen = {}
for cin, cout, k, s, h, w in conv_shapes(B, 256, 128):
    seen[(cin, cout, k, s, h, w)] = seen.get((cin, cout, k, s, h, w), 0) + 1
lib = L.lib()
tot = 0.0
for (cin, cout, k, s, h, w), cnt in seen.items():
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    pad = k // 2
    d, oh, ow = ly.conv_desc(B, h, w, cin, cout, k, s, pad)
    dy = torch.randn((B, oh, ow, cout), device="cuda").to(torch.bfloat16)
    nbytes = lib.creid_conv2d_wgrad_workspace_bytes(C.byref(d), L.BF16)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    def fn():
        L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(ws), nbytes, L.BF16, L.stream()), "p")
    t = time_kernel(fn, 10) * 1e3
    fl = 2.0 * B * oh * ow * cout * cin * k * k
    print(f"{cin:4d}->{cout:4d} k{k} s{s} M={B*oh*ow:6d} x{cnt}  {t:7.1f} us {fl/t/1e6:5.0f} TF/s  partials {nbytes/1e6:6.1f} MB")
    tot += t * cnt
print("per-step total (us):", round(tot))
