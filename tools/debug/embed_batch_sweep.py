import sys, json
sys.path.insert(0, ".")
import torch
from centroids_reid_amd import bench_train as bt
for B in (64, 128, 192, 256, 384, 512):
    r = bt.run_embed("resnet50", B, 256, 128, steps=10, warmup=3)
    print(B, round(r["value"]), "img/s", round(r["ms_per_step"], 3), "ms", flush=True)
