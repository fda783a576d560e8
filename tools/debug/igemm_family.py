"""One launch of every conv fwd + dgrad of the ResNet50 layer mix (B=64, 256x128) -- for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE) that back bench.py's roofline.traffic.  Also launches one l2norm over a known byte
count as the calibration kernel."""
import sys, torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, reid_metric as rm
from centroids_reid_amd.bench_train import conv_shapes
B = 64
alg = 0
for cin, cout, k, s, h, w in conv_shapes(B, 256, 128):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    y = ly.conv2d_fwd(x, krsc, s, k // 2, with_stats=True)[0]
    dx = ly.conv2d_dgrad(y, crsk, (h, w), s, k // 2)
    alg += 2 * (x.numel() + y.numel() + wt.numel()) * 2
f = torch.randn((20000, 2048), device="cuda")
rm.l2_normalize(f)
torch.cuda.synchronize()
print("algorithmic bytes (in+out+weights, fwd+dgrad):", alg, "calibration l2norm bytes:", f.numel() * 8)
