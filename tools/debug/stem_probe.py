"""Timing of the one-launch eval-mode stem (conv_stem.hip) against the two launches it replaces (run on the GPU box):
    python tools/debug/stem_probe.py [--B 128 --H 256 --W 128] [--abl]     (--abl: the ablation library, CREID_STEM_ABL sweeps)"""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=128)
ap.add_argument("--abl", action="store_true")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if args.abl:
    os.environ["CREID_LIB_PATH"] = os.path.join(ROOT, "centroids-reid_amd", "lib", "libcreid_hip_abl.so")
import torch
from centroids_reid_amd import _lib as L

lib = L.lib()
B, H, W = args.B, args.H, args.W
dtype = torch.bfloat16
xpad = torch.randn((B, H + 8, W + 6, 4), device="cuda").to(dtype)
w = (torch.randn((64, 256), device="cuda") / 12).to(dtype)
ss = torch.rand((2, 64), device="cuda") + 0.5
y0 = torch.empty((B, H // 2, W // 2, 64), device="cuda", dtype=dtype)
a = torch.empty((B, H // 4, W // 4, 64), device="cuda", dtype=dtype)
dt = L.dtype_code(a)
st = L.stream()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def two():
    L.check(lib.creid_stem_conv_fwd_affine(B, H, W, L.ptr(xpad), L.ptr(w), L.ptr(y0), L.ptr(ss), 1, dt, st), "stem")
    L.check(lib.creid_maxpool3x3s2_fwd(L.ptr(y0), B, H // 2, W // 2, 64, dt, L.ptr(a), None, st), "pool")


def one():
    L.check(lib.creid_stem_conv_pool_fwd_affine(B, H, W, L.ptr(xpad), L.ptr(w), L.ptr(a), L.ptr(ss), 1, dt, st), "stem_pool")


print(f"B={B} {H}x{W}: two launches {timeit(two):.1f} us", flush=True)
for form in ([0, 1] if W == 128 else [0]):
    os.environ["CREID_STEM_FORM"] = str(form)
    for wgs in (0, 128, 512):
        if wgs:
            os.environ["CREID_STEM_WGS"] = str(wgs)
        else:
            os.environ.pop("CREID_STEM_WGS", None)
        print(f"  one launch form {form} wgs {wgs or 'default'}: {timeit(one):.1f} us", flush=True)
        if args.abl and wgs == 0:
            for abl in (1, 2, 4, 8, 12, 13, 15):
                os.environ["CREID_STEM_ABL"] = str(abl)
                print(f"    abl {abl:2d} (1 no multiplies, 2 no loads, 4 no epilogue, 8 no pool): {timeit(one):.1f} us", flush=True)
            os.environ.pop("CREID_STEM_ABL", None)
os.environ.pop("CREID_STEM_WGS", None)
