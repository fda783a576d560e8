"""All-waves-multiply persistent igemm (conv_pipe.hip, igemm_bf16_pp_kernel) against the tile kernels: bit equality and time
per layer shape and variant.
    python tools/debug/pp_probe.py [batch [H W]] [--variants 0x...,0x...] [--min-k 256] [--modes stats,affine,res,dgrad]
variant word = BM / 128 | (BN / 128) << 2 | KPH << 4 | GLM << 8 (CREID_IGEMM_PP = 0x1000 | variant)."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly            # noqa: E402
from centroids_reid_amd.bench_train import conv_shapes  # noqa: E402
from bench import time_kernel                           # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = {a.split("=")[0][2:]: (a.split("=")[1] if "=" in a else "1") for a in sys.argv[1:] if a.startswith("--")}
B = int(args[0]) if len(args) > 0 else 64
H, W = (int(args[1]), int(args[2])) if len(args) > 2 else (256, 128)
MINK = int(opts.get("min-k", "256"))
MODES = opts.get("modes", "stats,affine,res,dgrad").split(",")


def vword(bm, bn, kph, glm):
    return (bm // 128) | ((bn // 128) << 2) | (kph << 4) | (glm << 8)


if "variants" in opts:
    VARIANTS = [int(v, 0) for v in opts["variants"].split(",")]
else:
    VARIANTS = [vword(bm, bn, kph, glm) for (bm, bn) in ((256, 256), (128, 256), (256, 128), (128, 128)) for (kph, glm) in ((1, 0), (2, 0), (1, 2))]


def vname(v):
    return f"{(v & 3) * 128}x{((v >> 2) & 3) * 128}k{(v >> 4) & 7}{'fmri'[(v >> 8) & 3]}"


def t_us(fn):
    return min(time_kernel(fn, 10) for _ in range(2)) * 1e3


seen = {}
for sh in conv_shapes(B, H, W):
    seen[sh] = seen.get(sh, 0) + 1
print(f"B={B} {H}x{W}: us per launch, tile kernels (rule/plan) vs pp variants [{' '.join(vname(v) for v in VARIANTS)}]; '!' = output differs")
tot = {m: [0.0, 0.0] for m in MODES}
for (cin, cout, k, s, h, w), cnt in seen.items():
    K = cin * k * k
    pad = k // 2
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
    y0 = ly.conv2d_fwd(x, krsc, s, pad)
    oh, ow = y0.shape[1], y0.shape[2]
    M = B * oh * ow
    res = torch.randn_like(y0)
    dy = torch.randn_like(y0)
    fns = {
        "stats": (lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), cout, K),
        "affine": (lambda: ly.conv2d_fwd_affine(x, krsc, s, pad, ss, None, True), cout, K),
        "res": (lambda: ly.conv2d_fwd_affine(x, krsc, s, pad, ss, res, True), cout, K),
        "dgrad": (lambda: ly.conv2d_dgrad(dy, crsk, (h, w), s, pad), cin, cout * k * k),
    }
    for mode in MODES:
        fn, N, Kg = fns[mode]
        if N % 128 or Kg < MINK:
            continue
        Mg = M if mode != "dgrad" else B * h * w
        os.environ.pop("CREID_IGEMM_PP", None)
        ref = fn()
        t0 = t_us(fn)
        cells, best, bestv = [], t0, None
        for v in VARIANTS:
            if N % (((v >> 2) & 3) * 128):
                cells.append("     -")
                continue
            os.environ["CREID_IGEMM_PP"] = hex(0x1000 | v)
            out = fn()
            if isinstance(ref, tuple):
                eq = torch.equal(out[0], ref[0]) and torch.equal(out[1].reshape(ref[1].shape), ref[1])
                close = torch.allclose(out[1], ref[1], rtol=1e-4, atol=1e-2) and torch.equal(out[0], ref[0])
            else:
                eq = close = torch.equal(out, ref)
            t = t_us(fn)
            cells.append(f"{t:6.1f}{'' if eq else ('~' if close else '!')}")
            if t < best:
                best, bestv = t, v
        os.environ.pop("CREID_IGEMM_PP", None)
        fl = 2.0 * Mg * N * Kg
        tot[mode][0] += t0 * cnt
        tot[mode][1] += best * cnt
        print(f"{mode:6s} {cin:4d}->{cout:4d} k{k} s{s} {h:3d}x{w:<3d} M={Mg:7d} N={N:4d} K={Kg:4d} x{cnt} tile {t0:6.1f} ({fl / t0 / 1e6:5.0f} TF) | "
              + " ".join(cells) + f" | best {vname(bestv) if bestv is not None else 'tile':>11s} {fl / best / 1e6:5.0f} TF", flush=True)
for m in MODES:
    print(f"sum {m}: tile {tot[m][0]:.0f} us, best-of {tot[m][1]:.0f} us")
