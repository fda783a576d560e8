"""A/B of environment switches on the captured eval-mode embedding forward (run on the GPU box):
    python tools/debug/embed_ab.py "CREID_C64_3X3=1" "CREID_C64_3X3=0" [--arch resnet50 --B 128 --H 256 --W 128]
Every configuration is captured afresh (the switches are read at launch = capture time), timed three times, interleaved twice."""
import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("cfgs", nargs="+")
ap.add_argument("--arch", default="resnet50")
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=128)
args = ap.parse_args()
import torch
from centroids_reid_amd.bench_train import EmbedBench

for rep in range(2):
    for cfg in args.cfgs:
        kv = dict(x.split("=", 1) for x in cfg.split())
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        eb = EmbedBench(args.arch, args.B, args.H, args.W)
        t = min(eb.run(30, 3) for _ in range(3))
        del eb
        torch.cuda.empty_cache()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
        print(f"{args.arch} B={args.B} {args.H}x{args.W} [{cfg}] : {t * 1e3:.4f} ms  ({args.B / t:.0f} img/s)", flush=True)
