import os, sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import layers as ly
B = 64
for cin, cout, h, w in ((64, 256, 64, 32), (256, 64, 64, 32), (128, 512, 32, 16)):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
    krsc, _ = ly.weight_prep(wt, torch.bfloat16)
    for tag, env in (("tile", {"CREID_STREAM1X1": "0"}), ("stream", {"CREID_STREAM1X1": "1"}), ("stream wgs128", {"CREID_STREAM1X1": "1", "CREID_STREAM1X1_WGS": "128"}),
                     ("stream wgs1024", {"CREID_STREAM1X1": "1", "CREID_STREAM1X1_WGS": "1024"})):
        os.environ.pop("CREID_STREAM1X1_WGS", None)
        os.environ.update(env)
        t1 = time_kernel(lambda: ly.conv2d_fwd(x, krsc, 1, 0, with_stats=True), 10) * 1e3
        t0 = time_kernel(lambda: ly.conv2d_fwd(x, krsc, 1, 0), 10) * 1e3
        print(f"{cin}->{cout} M={B*h*w} {tag:16s} with stats {t1:6.1f} us   without {t0:6.1f} us", flush=True)
