"""In-situ A/B of single launch plans on the captured embedding forward: the shipped plan file against the same file with one
entry dropped (or replaced), interleaved three times.
    python tools/debug/plan_ab.py --drop 1:16384,512,1024,10 [--set 1:16384,512,1024,10=128,3,3] [--arch resnet50 --B 128 --H 256 --W 128]"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--drop", action="append", default=[])
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--arch", default="resnet50")
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=128)
args = ap.parse_args()
import torch
from centroids_reid_amd import _lib as L
from centroids_reid_amd.bench_train import EmbedBench

plans = json.load(open(os.path.join(ROOT, "centroids-reid_amd", "tuned_plans.json")))["plans"]


def parse(s):
    kind, key = s.split(":")
    return int(kind), [int(v) for v in key.split(",")]


def variant(drop=None, setv=None):
    out = list(plans)
    if drop:
        k, key = parse(drop)
        out = [e for e in out if not (e["kind"] == k and e["key"] == key)]
    if setv:
        lhs, rhs = setv.split("=")
        k, key = parse(lhs)
        out = [e for e in out if not (e["kind"] == k and e["key"] == key)] + [dict(kind=k, key=key, plan=[int(v) for v in rhs.split(",")])]
    return out


def time_with(entries):
    L.lib().creid_tune_clear()
    f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, dir="/tmp")
    json.dump({"plans": entries}, f); f.close()
    L.load_tuned_plans(f.name); os.unlink(f.name)
    eb = EmbedBench(args.arch, args.B, args.H, args.W)
    t = min(eb.run(30, 3) for _ in range(3))
    del eb
    torch.cuda.empty_cache()
    return t * 1e3


cfgs = [("shipped", plans)] + [(f"drop {d}", variant(drop=d)) for d in args.drop] + [(f"set {s}", variant(setv=s)) for s in args.set]
for rep in range(3):
    for name, ents in cfgs:
        print(f"{name}: {time_with(ents):.4f} ms", flush=True)
