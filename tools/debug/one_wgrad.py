"""Run one conv wgrad shape repeatedly (for rocprofv3 --pmc): python tools/debug/one_wgrad.py cin cout k stride h w [reps]"""
import sys, torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly
cin, cout, k, s, h, w = map(int, sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
B = 64
x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
y = ly.conv2d_fwd(x, krsc, s, k // 2)
dw = torch.zeros((cout, cin, k, k), device="cuda")
for _ in range(reps):
    ly.conv2d_wgrad(x, y, k, s, k // 2, out=dw)
torch.cuda.synchronize()
