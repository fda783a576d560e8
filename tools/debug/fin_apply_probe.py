"""Training-mode BatchNorm: finalize + apply as two launches (shipped until round 5) against creid_bn2d_finalize_apply_mask (one
launch, every apply workgroup sums the partial rows of its own channel strip) on the shapes of the B = 64 training step that have
<= 256 statistic rows -- us per layer for both forms (10 per graph, warm operands), the row-block count swept, and the outputs
compared BIT FOR BIT (y, ReLU bits, mean, invstd, scale_shift, running statistics).
    python tools/debug/fin_apply_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import _lib as L   # noqa: E402
from bench import time_kernel              # noqa: E402

lib = L.lib()
dt = torch.bfloat16
print("M x C (+res): two launches us | fused us at row_blocks = auto, 4, 8, 16, 32, 64 | bit-identical")
tot2 = totf = 0.0
for M, C, with_res, cnt in ((32768, 128, False, 7), (32768, 512, True, 3), (32768, 256, False, 1), (8192, 256, False, 11),
                            (8192, 1024, True, 5), (8192, 512, False, 6), (8192, 2048, True, 2)):
    rows = (M + 127) // 128
    x = torch.randn((M, C), device="cuda").to(dt)
    res = torch.randn((M, C), device="cuda").to(dt) if with_res else None
    xf = x.float().view(rows, -1, C)
    part = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()          # [rows][2][C]
    gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda") * 0.1

    def fresh():
        return dict(rm=torch.randn(C, device="cuda") * 0.1, rv=torch.rand(C, device="cuda") + 0.5, mean=torch.empty(C, device="cuda"),
                    inv=torch.empty(C, device="cuda"), ss=torch.empty((2, C), device="cuda"), y=torch.empty_like(x),
                    mask=torch.empty(M * C // 8, dtype=torch.uint8, device="cuda"))
    torch.manual_seed(0); a = fresh(); torch.manual_seed(0); b = fresh()

    def two(o=a):
        L.check(lib.creid_bn2d_finalize(L.ptr(part), rows, C, M, L.ptr(o["rm"]), L.ptr(o["rv"]), 1, 0.1, 1e-5, L.ptr(gamma), L.ptr(beta),
                                        L.ptr(o["mean"]), L.ptr(o["inv"]), L.ptr(o["ss"]), L.stream()), "fin")
        L.check(lib.creid_bn2d_apply_mask(L.ptr(x), L.ptr(o["ss"]), L.ptr(res), 1, M, C, L.BF16, L.ptr(o["y"]), L.ptr(o["mask"]),
                                          L.stream()), "apply")

    def fused(rb=0, o=b):
        L.check(lib.creid_bn2d_finalize_apply_mask(L.ptr(part), rows, C, M, L.ptr(o["rm"]), L.ptr(o["rv"]), 0.1, 1e-5, L.ptr(gamma),
                                                   L.ptr(beta), L.ptr(o["mean"]), L.ptr(o["inv"]), L.ptr(o["ss"]), L.ptr(x), L.ptr(res),
                                                   1, L.BF16, L.ptr(o["y"]), L.ptr(o["mask"]), rb, L.stream()), "fused")
    two(); fused(); torch.cuda.synchronize()
    same = all(torch.equal(a[k], b[k]) for k in a)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    t2 = min(time_kernel(two, 10) for _ in range(2)) * 1e3
    tf = [min(time_kernel(lambda rb=rb: fused(rb), 10) for _ in range(2)) * 1e3 for rb in (0, 4, 8, 16, 32, 64)]
    tot2 += t2 * cnt; totf += tf[0] * cnt
    print(f"{M:6d} x {C:4d} {'+res' if with_res else '    '} x{cnt:2d}: {t2:6.1f} | " + " ".join(f"{t:6.1f}" for t in tf) +
          f" | {'identical' if same else 'DIFFERENT: ' + ','.join(bad)}", flush=True)
print(f"per step over these {35} launches: two launches {tot2:.0f} us, fused (auto) {totf:.0f} us")
