"""Why is a bottleneck's conv1 data gradient 2-3 x its same-shape forward inside the training step (profiles/r05_train_layers.md)?
The launch computes dX = dY . W (M x planes -> M x cin: a short-K, wide-N GEMM) and its copy-out also (a) adds the block's
incoming gradient through the ReLU bit mask (the residual branch) and (b) carries the column reduction of the PREVIOUS block's
bn3 backward (reads that layer's raw output x3 and its mask once more).  Per shape: the plain data gradient, + (a), + (b),
+ both (the shipped launch), and the forward convolution of the mirrored GEMM (conv3: planes -> cin with BatchNorm partial sums)
-- isolated launches on warm operands, 10 per graph.
    python tools/debug/dgrad_c1_probe.py [batch]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L   # noqa: E402
from bench import time_kernel                             # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
print(f"B={B}: conv1 data gradient (cin <- planes) in us: plain | + masked residual add | + bn3 column sums | + both (shipped) || "
      "mirrored forward conv3 with BN partials; bytes of the shipped launch; its rate")
for cin, pl, h, w in ((256, 64, 64, 32), (512, 128, 32, 16), (1024, 256, 16, 8), (2048, 512, 16, 8)):
    M = B * h * w
    d, _, _ = ly.conv_desc(B, h, w, cin, pl, 1, 1, 0)
    dy = torch.randn((B, h, w, pl), device="cuda").to(torch.bfloat16)
    wt = torch.randn((pl, cin, 1, 1), device="cuda") / cin ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    dx = torch.empty((M, cin), dtype=torch.bfloat16, device="cuda")
    add = torch.randn((M, cin), device="cuda").to(torch.bfloat16)
    x3 = torch.randn((M, cin), device="cuda").to(torch.bfloat16)
    mask = torch.randint(0, 256, (M * cin // 8,), dtype=torch.uint8, device="cuda")
    mean = torch.zeros(cin, device="cuda"); invstd = torch.ones(cin, device="cuda")
    part = torch.empty((lib.creid_bn2d_bwd_rows(M) * 2, cin), dtype=torch.float32, device="cuda")

    def run(with_add, with_bn):
        def f():
            L.check(lib.creid_conv2d_dgrad_fused_nhwc(C.byref(d), L.ptr(dy), L.ptr(crsk), L.ptr(dx), L.ptr(add) if with_add else None, 1,
                                                      L.ptr(mask) if with_add else None, L.ptr(x3) if with_bn else None, None,
                                                      L.ptr(mask) if with_bn else None, L.ptr(mean) if with_bn else None,
                                                      L.ptr(invstd) if with_bn else None, L.ptr(part) if with_bn else None, 0, None, None, 1,
                                                      None, 0, L.BF16, L.stream()), "dgrad")
        return time_kernel(f, 10) * 1e3
    t = [run(False, False), run(True, False), run(False, True), run(True, True)]
    # mirrored forward: conv3 planes -> cin on the same pixel grid, BatchNorm partial sums in the epilogue
    d3, _, _ = ly.conv_desc(B, h, w, pl, cin, 1, 1, 0)
    w3 = torch.randn((cin, pl, 1, 1), device="cuda") / pl ** 0.5
    k3, _ = ly.weight_prep(w3, torch.bfloat16)
    a2 = torch.randn((B, h, w, pl), device="cuda").to(torch.bfloat16)
    y3 = torch.empty((M, cin), dtype=torch.bfloat16, device="cuda")
    p3 = torch.empty((lib.creid_conv2d_bn_partial_rows(C.byref(d3)) * 2, cin), dtype=torch.float32, device="cuda")
    tf = time_kernel(lambda: L.check(lib.creid_conv2d_fwd_nhwc(C.byref(d3), L.ptr(a2), L.ptr(k3), L.ptr(y3), L.ptr(p3), L.BF16, L.stream()),
                                     "fwd"), 10) * 1e3
    by = M * (pl + 3 * cin) * 2 + M * cin // 4
    print(f"{cin:4d}<-{pl:<3d} {h:2d}x{w:<2d} M={M:6d}: {t[0]:6.1f} | {t[1]:6.1f} | {t[2]:6.1f} | {t[3]:6.1f} || fwd {tf:6.1f}; "
          f"{by / 1e6:6.1f} MB; {by / t[3] / 1e6:5.2f} TB/s; separate passes instead of fusing: add {M * cin * 6 / 6.0e6:5.1f} us + reduce "
          f"{M * cin * 4.25 / 6.0e6:5.1f} us at 6 TB/s", flush=True)
