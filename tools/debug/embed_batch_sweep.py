"""Eval-mode embedding forward against the batch size (hipGraph replay):
    python tools/debug/embed_batch_sweep.py [arch H W batch ...]      default: resnet50 256 128 64 128 192 256 384 512"""
import sys
sys.path.insert(0, ".")
from centroids_reid_amd import bench_train as bt   # noqa: E402

a = sys.argv[1:]
arch, H, W = (a[0], int(a[1]), int(a[2])) if len(a) >= 3 else ("resnet50", 256, 128)
batches = [int(v) for v in a[3:]] or [64, 128, 192, 256, 384, 512]
for B in batches:
    r = bt.run_embed(arch, B, H, W, steps=10 if B * H * W < 40e6 else 4, warmup=3)
    print(arch, f"{H}x{W}", B, round(r["value"]), "img/s", round(r["ms_per_step"], 3), "ms", flush=True)
