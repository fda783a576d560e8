"""A few replays of the captured eval-mode embedding forward of any configuration, for a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o e -- python tools/debug/embed_trace.py --arch resnet50_ibn_a --B 256 --H 320 --W 320"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="resnet50")
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=128)
args = ap.parse_args()
import torch
from centroids_reid_amd.bench_train import EmbedBench

eb = EmbedBench(args.arch, args.B, args.H, args.W)
print(f"{eb.run(5, 2) * 1e3:.3f} ms per forward")
torch.cuda.synchronize()
