"""How far are the 1x1 igemm shapes from the vendor GEMM?  torch.mm (hipBLASLt / rocBLAS) on the same M, N, K in bf16,
un-profiled graph replays -- a yardstick only, the product never calls it."""
import sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import layers as ly
shapes = [(131072, 64, 64), (131072, 256, 64), (131072, 64, 256), (32768, 512, 128), (32768, 128, 512), (8192, 1024, 256),
          (8192, 256, 1024), (8192, 2048, 512), (8192, 512, 2048), (8192, 2048, 1024), (8192, 512, 4608), (8192, 256, 2304)]
for M, N, K in shapes:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    b = torch.randn((N, K), device="cuda").to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    t = time_kernel(lambda: torch.mm(a, b.t(), out=out), 10) * 1e3
    line = f"M={M:6d} N={N:4d} K={K:4d}  torch.mm {t:6.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s"
    if K <= 2048:
        x = a.view(64, -1, 1, K) if False else a.view(64, M // 64, 1, K)
        w = torch.randn((N, K, 1, 1), device="cuda") / K ** 0.5
        krsc, _ = ly.weight_prep(w, torch.bfloat16)
        t2 = time_kernel(lambda: ly.conv2d_fwd(x, krsc, 1, 0), 10) * 1e3
        line += f" | igemm 1x1 {t2:6.1f} us {2.0*M*N*K/t2/1e6:6.0f} TF/s"
    print(line, flush=True)
# wgrad-shaped: dW[N][K] = dY[M][N]^T X[M][K], contraction over M = 8192 pixels
for N, K in ((2048, 512), (512, 2048), (1024, 256), (512, 4608)):
    M = 8192
    dy = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    out = torch.empty((N, K), device="cuda", dtype=torch.bfloat16)
    t = time_kernel(lambda: torch.mm(dy.t(), x, out=out), 10) * 1e3
    print(f"wgrad-shaped N={N} K={K} over {M} px: torch.mm {t:6.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
