"""Workload of the embedding-forward counter passes (tools/pmc_run.sh): three eval-mode forwards of ResNet50 at batch 128, eager
(no graph: every kernel is a dispatch the profiler sees)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centroids_reid_amd.bench_train import make_model      # noqa: E402

model = make_model()
model.eval()
x = torch.randn((128, 3, 256, 128), device="cuda")
with torch.no_grad():
    for _ in range(3):
        _, f = model.backbone(x)
        model.bn(f)
torch.cuda.synchronize()
