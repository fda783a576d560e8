"""Run a few representative ResNet50 conv shapes through the bf16 implicit-GEMM kernel (for rocprofv3 --pmc)."""
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
B = 64
shapes = [(64, 64, 3, 1, 64, 32), (256, 256, 3, 1, 16, 8), (512, 512, 3, 1, 16, 8), (1024, 256, 1, 1, 16, 8), (64, 256, 1, 1, 64, 32)]
for cin, cout, k, s, h, w in shapes:
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    for _ in range(5):
        y = ly.conv2d_fwd(x, krsc, s, k // 2, with_stats=True)
torch.cuda.synchronize()
print("done")
