"""us per call of the one-launch weight preparation (fp32 OIHW -> 16-bit [O][r][s][I] + [I][r][s][O]) over ResNet50's 52 non-stem
convolutions, as the training step issues it."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from centroids_reid_amd import _lib as L
from centroids_reid_amd.bench_train import conv_shapes
lib = L.lib()
shapes = [(cout, cin, k) for cin, cout, k, s, h, w in conv_shapes(64, 256, 128)]
ws = [torch.randn((o, i, k, k), device="cuda") for o, i, k in shapes]
krsc = [torch.empty((o, k, k, i), device="cuda", dtype=torch.bfloat16) for o, i, k in shapes]
crsk = [torch.empty((i, k, k, o), device="cuda", dtype=torch.bfloat16) for o, i, k in shapes]
rec = np.zeros(len(shapes), dtype=np.dtype([("w", "<u8"), ("krsc", "<u8"), ("crsk", "<u8"), ("O", "<i4"), ("I", "<i4"),
                                             ("kh", "<i4"), ("kw", "<i4"), ("start", "<i8")]))
start, tiles, tstart = 0, 0, np.zeros(len(shapes), np.int32)
for n, (o, i, k) in enumerate(shapes):
    rec[n] = (ws[n].data_ptr(), krsc[n].data_ptr(), crsk[n].data_ptr(), o, i, k, k, start)
    start += o * i * k * k
    tstart[n] = tiles
    tiles += ((o + 31) // 32) * ((i + 31) // 32)
tab, ts = torch.from_numpy(rec.view(np.uint8).copy()).cuda(), torch.from_numpy(tstart).cuda()
dt, st = L.dtype_code(krsc[0]), L.stream()
f = lambda: L.check(lib.creid_weight_prep_multi(L.ptr(tab), L.ptr(ts), len(shapes), tiles, dt, st), "multi")
for _ in range(3):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{len(shapes)} tensors, {start / 1e6:.1f} M weights, {tiles} tiles: {us:.1f} us  ({start * 8 / us / 1e3:.0f} GB/s)")
