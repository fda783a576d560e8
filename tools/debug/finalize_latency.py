"""Per-launch cost of the BN finalize kernels inside a captured chain (graph replay of 50 dependent-in-order launches),
next to a one-element torch kernel as the launch floor."""
import sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import _lib as L
lib = L.lib()
st = L.stream
one = torch.zeros(1, device="cuda")
t0 = time_kernel(lambda: one.add_(1.0), 50) * 1e3
print(f"one-element torch kernel: {t0:.2f} us per launch")
for rows, C in ((1024, 64), (1024, 256), (256, 128), (256, 512), (64, 256), (64, 1024), (64, 512), (64, 2048)):
    part = torch.rand((rows, 2, C), device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, inv, ss = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty((2, C), device="cuda")
    f = lambda: L.check(lib.creid_bn2d_finalize(L.ptr(part), rows, C, rows * 128, L.ptr(rm), L.ptr(rv), 1, 0.1, 1e-5, L.ptr(g),
                                                L.ptr(b), L.ptr(mean), L.ptr(inv), L.ptr(ss), st()), "fin")
    t = time_kernel(f, 50) * 1e3
    print(f"bn2d_finalize rows={rows:5d} C={C:5d}: {t:.2f} us per launch", flush=True)
