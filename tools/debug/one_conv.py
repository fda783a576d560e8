"""Run one conv shape repeatedly (for rocprofv3 --pmc): python tools/debug/one_conv.py cin cout k stride h w [reps]"""
import sys, torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
cin, cout, k, s, h, w = map(int, sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
B = 64
x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
for _ in range(reps):
    y = ly.conv2d_fwd(x, krsc, s, k // 2, with_stats=True)
torch.cuda.synchronize()
