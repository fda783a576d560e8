"""256-row igemm tiles (igemm_bf16_ws_kernel<128, NS, 256>) against the 128-row tiles: bit equality and time per shape.
    python tools/debug/bm256_probe.py [batch]"""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L   # noqa: E402
from centroids_reid_amd.bench_train import conv_shapes    # noqa: E402
from bench import time_kernel                             # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 128)
os.environ["CREID_IGEMM_BM256_MIN_WGS"] = "1"
seen = {}
for sh in conv_shapes(B, H, W):
    seen[sh] = seen.get(sh, 0) + 1
print(f"B={B} {H}x{W}: forward with folded BN epilogue; us(128-row rule/plan) us(256-row ns3) ; equal ; TF/s best")
tot128 = tot256 = totbest = 0.0
for (cin, cout, k, s, h, w), cnt in seen.items():
    if cout % 128:
        continue
    pad = k // 2
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
    res = {}
    r = None
    for bm in ("128", "256"):
        os.environ["CREID_IGEMM_BM"] = bm
        y = ly.conv2d_fwd_affine(x, krsc, s, pad, ss, None, True)
        ys, part = ly.conv2d_fwd(x, krsc, s, pad, with_stats=True)
        if r is None:
            r = torch.randn_like(y)
        yr = ly.conv2d_fwd_affine(x, krsc, s, pad, ss, r, True)
        t = time_kernel(lambda: ly.conv2d_fwd_affine(x, krsc, s, pad, ss, None, True), 10) * 1e3
        t = min(t, time_kernel(lambda: ly.conv2d_fwd_affine(x, krsc, s, pad, ss, None, True), 10) * 1e3)
        ts = time_kernel(lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), 10) * 1e3
        dx = ly.conv2d_dgrad(ys, crsk, (h, w), s, pad) if cin % 128 == 0 else None
        res[bm] = (y, ys, part.sum(0), yr, t, ts, dx)
    os.environ.pop("CREID_IGEMM_BM")
    a, b = res["128"], res["256"]
    eq = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    eqp = bool(torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-3))
    eqd = a[6] is None or torch.equal(a[6], b[6])
    fl = 2.0 * B * a[0].shape[1] * a[0].shape[2] * cout * cin * k * k
    best = min(a[4], b[4])
    tot128 += a[4] * cnt; tot256 += b[4] * cnt; totbest += best * cnt
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {h:3d}x{w:<3d} M={B*a[0].shape[1]*a[0].shape[2]:6d} x{cnt}  affine {a[4]:7.1f} {b[4]:7.1f}  stats {a[5]:7.1f} {b[5]:7.1f}"
          f"  eq={eq} part={eqp} dgrad={eqd}  {fl/best/1e6:5.0f} TF/s {'<-256' if b[4] < 0.97 * a[4] else ''}")
print(f"sum over N%128==0 convs (us): 128-row {tot128:.0f}  256-row {tot256:.0f}  best-of {totbest:.0f}")
