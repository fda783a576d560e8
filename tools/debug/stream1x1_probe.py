import os as _os; _os.environ.setdefault("CREID_DEBUG_KNOBS", "1")   # the CREID_* knobs below are flipped inside this process (csrc/common.hpp)
import os, sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import layers as ly
B = 64
for cin, cout, h, w in ((64, 256, 64, 32),):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5
    krsc, _ = ly.weight_prep(wt, torch.bfloat16)
    os.environ["CREID_STREAM1X1"] = "1"
    for dbg in (0, 1, 2, 3, 4, 7):
        os.environ["CREID_STREAM1X1_DBG"] = str(dbg)
        t0 = time_kernel(lambda: ly.conv2d_fwd(x, krsc, 1, 0), 10) * 1e3
        print(f"{cin}->{cout} dbg={dbg} (1 = no global stores, 2 = no staging, 4 = no mfma): {t0:6.1f} us", flush=True)
