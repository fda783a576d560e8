"""Timing ablation and cycle trace of the all-waves-multiply igemm (conv_pipe.hip) on a few layer shapes (ablation build:
python centroids-reid_amd/build.py --ablation).  CREID_IGEMM_ABL bits: 1 no MFMA, 2 no DMA after the first k-tile, 4 no fragment
reads, 8 no copy-out stores.
    python tools/debug/pp_abl.py [batch] [--variants=0x..,..] [--trace]"""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import ctypes as C
import os
import sys
import torch
os.environ.setdefault("CREID_LIB_PATH", "centroids-reid_amd/lib/libcreid_hip_abl.so")
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L    # noqa: E402
from bench import time_kernel                              # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = {a.split("=")[0][2:]: (a.split("=")[1] if "=" in a else "1") for a in sys.argv[1:] if a.startswith("--")}
B = int(args[0]) if args else 128
VARIANTS = [int(v, 0) for v in opts.get("variants", "0x1a,0x21a,0x29,0x219").split(",")]
SHAPES = [(512, 512, 3, 1, 16, 8), (1024, 2048, 1, 1, 16, 8), (512, 2048, 1, 1, 16, 8), (2048, 512, 1, 1, 16, 8), (256, 1024, 1, 1, 16, 8)]
ABLS = [0, 8, 2, 10, 1, 3, 11]


def vname(v):
    return f"{(v & 3) * 128}x{((v >> 2) & 3) * 128}k{(v >> 4) & 7}{'fmri'[(v >> 8) & 3]}"


def t_us(fn):
    return min(time_kernel(fn, 10) for _ in range(2)) * 1e3


lib = L.lib()
print(f"B={B}: forward with the statistics epilogue, us per launch; columns = CREID_IGEMM_ABL {ABLS} (1 MFMA, 2 DMA, 4 reads, 8 stores removed)")
for (cin, cout, k, s, h, w) in SHAPES:
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    pad = k // 2
    fn = lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True)
    os.environ.pop("CREID_IGEMM_PP", None)
    os.environ["CREID_IGEMM_ABL"] = "0"
    base = t_us(fn)
    for v in VARIANTS:
        if cout % (((v >> 2) & 3) * 128):
            continue
        os.environ["CREID_IGEMM_PP"] = hex(0x1000 | v)
        cells = []
        for a in ABLS:
            os.environ["CREID_IGEMM_ABL"] = str(a)
            cells.append(f"{t_us(fn):6.1f}")
        os.environ["CREID_IGEMM_ABL"] = "0"
        print(f"{cin:4d}->{cout:4d} k{k} M={B * h * w:6d} tile {base:6.1f} | {vname(v):>11s} " + " ".join(cells), flush=True)
        if "trace" in opts:
            n = 1024
            buf = torch.zeros((2, n), dtype=torch.int64, device="cuda")
            lib.creid_dbg_pp_trace.argtypes = [C.c_void_p]
            lib.creid_dbg_pp_trace(C.c_void_p(buf.data_ptr()))
            fn()
            torch.cuda.synchronize()
            lib.creid_dbg_pp_trace(C.c_void_p(0))
            tr = buf.cpu().numpy()
            for g in range(2):
                st = tr[g][tr[g] > 0]
                d = (st[1:] - st[:-1]).tolist()
                print(f"   trace group {g}: {len(st)} stamps; first 44 deltas (cycles): {d[:44]}")
                print(f"                  last 16 deltas: {d[-16:]}  total {int(st[-1] - st[0])}")
            off = tr[1][0] - tr[0][0]
            print(f"   group 1 first stamp - group 0 first stamp = {int(off)}")
    os.environ.pop("CREID_IGEMM_PP", None)
