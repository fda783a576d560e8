import sys
sys.path.insert(0, ".")
import torch
from bench import time_kernel
from centroids_reid_amd import _lib as L
lib = L.lib(); dt = L.BF16
B, H, W, Cc = 64, 128, 64, 64
M = B * H * W
x = torch.randn((M, Cc), device="cuda").to(torch.bfloat16)
ss = torch.stack([torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.3]).contiguous()
y = torch.empty_like(x)
p = torch.empty((M // 4, Cc), device="cuda", dtype=torch.bfloat16); idx = torch.empty((M // 4, Cc), device="cuda", dtype=torch.uint8)
st = L.stream
t_apply = time_kernel(lambda: L.check(lib.creid_bn2d_apply(L.ptr(x), L.ptr(ss), None, 0, M, Cc, dt, L.ptr(y), st()), "a"), 10) * 1e3
t_pool = time_kernel(lambda: L.check(lib.creid_maxpool3x3s2_fwd(L.ptr(y), B, H, W, Cc, dt, L.ptr(p), L.ptr(idx), st()), "p"), 10) * 1e3
t_fused = time_kernel(lambda: L.check(lib.creid_bn2d_apply_maxpool3x3s2(L.ptr(x), L.ptr(ss), 0, B, H, W, Cc, dt, L.ptr(p), L.ptr(idx), st()), "f"), 10) * 1e3
print(f"fwd: apply {t_apply:.1f} + maxpool {t_pool:.1f} us   vs fused {t_fused:.1f} us")
g = torch.randn((M // 4, Cc), device="cuda").to(torch.bfloat16)
mean, invstd, gamma = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda"), torch.ones(Cc, device="cuda")
rows = lib.creid_bn2d_bwd_rows(M)
part = torch.empty((rows, 2, Cc), device="cuda"); sums = torch.empty((3, Cc), device="cuda")
dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
dy, dx = torch.empty_like(x), torch.empty_like(x)
t_pb = time_kernel(lambda: L.check(lib.creid_maxpool3x3s2_bwd(L.ptr(g), L.ptr(idx), B, H, W, Cc, dt, L.ptr(dy), st()), "pb"), 10) * 1e3
t_bn = time_kernel(lambda: L.check(lib.creid_bn2d_bwd(L.ptr(x), L.ptr(dy), None, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), M, Cc, dt, L.ptr(part), 0,
                                                      L.ptr(sums), L.ptr(dg), L.ptr(db), L.ptr(dx), None, st()), "b"), 10) * 1e3
t_fb = time_kernel(lambda: L.check(lib.creid_bn2d_bwd_pooled(L.ptr(x), L.ptr(g), L.ptr(idx), B, H, W, None, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), Cc, dt,
                                                             L.ptr(part), L.ptr(sums), L.ptr(dg), L.ptr(db), L.ptr(dx), st()), "fb"), 10) * 1e3
print(f"bwd: maxpool_bwd {t_pb:.1f} + bn_bwd(reduce+finalize+apply) {t_bn:.1f} us   vs pooled {t_fb:.1f} us")
