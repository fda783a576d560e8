"""Timing sweep of the streamed count kernel (run on the GPU box): python tools/debug/stream_sweep.py"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from bench import time_kernel, eval_inputs
from centroids_reid_amd import reid_metric as rm
L = rm.L
nq, ng, D = 2228, 17661, 2048
feats, pids, cams = eval_inputs(nq, ng, D, 0, 1)
plan = rm.StreamPlan(pids[:nq], pids[nq:], cams[:nq], cams[nq:], "cuda")
fn, sq = rm.l2_normalize(feats, return_sqnorm=True)
q, g = fn[:nq], fn[nq:]; qq, gg = sq[:nq].contiguous(), sq[nq:].contiguous()
cap = plan.cap
lib = L.lib()
pos_key = torch.empty((nq, cap), dtype=torch.int32, device="cuda"); pos_idx = torch.empty_like(pos_key)
npos = torch.empty(nq, dtype=torch.int32, device="cuda"); hist = torch.zeros((nq, cap), dtype=torch.int32, device="cuda")
L.check(lib.creid_stream_poslist(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_slot), L.ptr(plan.csr_off),
                                 L.ptr(plan.g_order), L.ptr(plan.q_cams), L.ptr(plan.g_cams), cap, L.ptr(pos_key), L.ptr(pos_idx),
                                 L.ptr(npos), L.stream()), "poslist")
def count():
    L.check(lib.creid_stream_count(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_pids), L.ptr(plan.g_pids), cap,
                                   L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), L.ptr(hist), L.stream()), "count")
t = time_kernel(count, 10)
fl = 2.0 * nq * ng * D
print(f"cap {cap} count {t*1e3:.1f} us  {fl/t/1e9:.1f} TF/s  {fl/t/1e9/157.3:.3f} of peak")
