"""Workload for the rocprofv3 --pmc passes (run on the GPU box through tools/pmc_run.sh): ONE launch of every kernel the
roofline report talks about, plus calibration kernels whose HBM byte counts are known exactly.  Prints the algorithmic
byte / flop counts as JSON (first line starting with PMCMETA)."""
import json
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centroids_reid_amd import layers as ly, reid_metric as rm, _lib as L      # noqa: E402
from centroids_reid_amd.bench_train import conv_shapes                          # noqa: E402
from bench import eval_inputs                                                   # noqa: E402

meta = {}
B = 64
# ---- calibration: streaming kernels with exactly known traffic, buffers far larger than L2 and not MALL-resident
f = torch.randn((20000, 2048), device="cuda")                                   # 163.84 MB fp32
torch.cuda.synchronize()
rm.l2_normalize(f)                                                              # global_load_dwordx4 read + write
meta["calib_l2norm_rows"] = {"read": f.numel() * 4, "write": f.numel() * 4, "kernel": "l2norm_rows_kernel"}
M, C = 131072 * 4, 256
xb = torch.randn((M, C), device="cuda").to(torch.bfloat16); rb = torch.randn((M, C), device="cuda").to(torch.bfloat16)
ss = torch.ones((2, C), device="cuda"); ob = torch.empty_like(xb)
L.check(L.lib().creid_bn2d_apply(L.ptr(xb), L.ptr(ss), L.ptr(rb), 1, M, C, L.BF16, L.ptr(ob), L.stream()), "apply")
meta["calib_bn2d_apply"] = {"read": 2 * xb.numel() * 2, "write": xb.numel() * 2, "kernel": "bn2d_apply_kernel"}
del xb, rb, ob
# LDS-DMA streaming: a 1x1 64 -> 64 convolution over 2M pixels reads every input byte exactly once through
# global_load_lds (the 8 KB of weights stay in L2) and writes every output byte once
xc = torch.randn((B * 8, 64, 64, 64), device="cuda").to(torch.bfloat16)         # [512,64,64,64] NHWC = 268 MB
wt = torch.randn((64, 64, 1, 1), device="cuda") / 8
krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
yc = ly.conv2d_fwd(xc, krsc, 1, 0)
meta["calib_igemm_1x1_stream"] = {"read": xc.numel() * 2, "write": yc.numel() * 2, "kernel": "igemm_bf16_ws_kernel<64, 2, 128>",
                                  "note": "the ONLY launch of this template instance before the family below starts; dispatch order"}
del xc, yc
torch.cuda.synchronize()
print("PMCMARK calibration_done", flush=True)
# ---- igemm / wgrad families over the real layer mix
alg_ig = fl_ig = alg_wg = fl_wg = 0
for cin, cout, k, s, h, w in conv_shapes(B, 256, 128):
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    y = ly.conv2d_fwd(x, krsc, s, k // 2, with_stats=True)[0]
    ly.conv2d_dgrad(y, crsk, (h, w), s, k // 2)
    dw = torch.zeros((cout, cin, k, k), device="cuda")
    ly.conv2d_wgrad(x, y, k, s, k // 2, out=dw)
    fl = 2.0 * B * y.shape[1] * y.shape[2] * cout * cin * k * k
    alg_ig += 2 * (x.numel() + y.numel() + wt.numel()) * 2; fl_ig += 2 * fl
    alg_wg += (x.numel() + y.numel()) * 2 + wt.numel() * 4; fl_wg += fl
meta["igemm_family"] = {"algorithmic_bytes": alg_ig, "flops": fl_ig, "launches": 104, "kernel_prefix": "igemm_bf16_"}
meta["wgrad_family"] = {"algorithmic_bytes": alg_wg, "flops": fl_wg, "launches": 52, "kernel_prefix": "wgrad_"}
# ---- evaluation kernels (BASELINE configs[4])
nq, ng, D = 2228, 17661, 2048
feats, pids, cams = eval_inputs(nq, ng, D, 0, 1)
fn, sq = rm.l2_normalize(feats, return_sqnorm=True)
q, g = fn[:nq], fn[nq:]; qq, gg = sq[:nq].contiguous(), sq[nq:].contiguous()
d = rm.get_euclidean(q, g, qq, gg)
idx = rm.rank_rows(d)
rm.eval_func_device(idx, pids[:nq], pids[nq:], cams[:nq], cams[nq:], 50)
plan = rm.StreamPlan(pids[:nq], pids[nq:], cams[:nq], cams[nq:], "cuda")
rm.stream_eval(q, g, qq, gg, plan)
fl = 2.0 * nq * ng * D
meta["sqdist_f32_kernel"] = {"algorithmic_bytes": (nq + ng) * D * 4 + nq * ng * 4, "flops": fl, "launches": 1}
meta["sqdist_count_f32_kernel"] = {"algorithmic_bytes": (nq + ng) * D * 4 + 2 * nq * plan.cap * 4, "flops": fl, "launches": 1}
meta["rank_rows_lds_kernel"] = {"algorithmic_bytes": nq * ng * 12, "launches": 1}
torch.cuda.synchronize()
print("PMCMETA " + json.dumps(meta), flush=True)
