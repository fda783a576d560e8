"""Time creid_stream_count (the streamed evaluation's contraction + counting kernel) alone, Duke-shaped problem; with
CREID_STREAM_NOEPI=1 the counting epilogue is skipped (timing only).  Run on the GPU box."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroids_reid_amd import reid_metric as rm, _lib as L
from bench import time_kernel
nq, ng, D = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2048) if len(sys.argv) > 2 else (2228, 17661, 2048)
rng = np.random.default_rng(0)
feats = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32)).cuda()
pids = np.concatenate([rng.integers(0, 702, nq), rng.integers(0, 1110, ng)]); cams = rng.integers(0, 8, nq + ng)
fn, sq = rm.l2_normalize(feats, return_sqnorm=True)
q, g, qq, gg = fn[:nq], fn[nq:], sq[:nq], sq[nq:]
plan = rm.StreamPlan.on_device(pids, cams, nq, "cuda").finish()
cap = plan.cap
lib = L.lib()
pos_key = torch.empty((nq, cap), dtype=torch.int32, device="cuda"); pos_idx = torch.empty_like(pos_key)
npos = torch.empty(nq, dtype=torch.int32, device="cuda"); hist = torch.zeros((nq, cap), dtype=torch.int32, device="cuda")
L.check(lib.creid_stream_poslist(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_slot), L.ptr(plan.csr_off),
                                 L.ptr(plan.g_order), L.ptr(plan.q_cams), L.ptr(plan.g_cams), cap, L.ptr(pos_key), L.ptr(pos_idx),
                                 L.ptr(npos), L.stream()), "poslist")
count = lambda: L.check(lib.creid_stream_count(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_pids),
                                               L.ptr(plan.g_pids), cap, L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), L.ptr(hist),
                                               L.stream()), "count")
t = min(time_kernel(count, 10) for _ in range(2))
td = min(time_kernel(lambda: rm.get_euclidean(q, g, qq, gg), 10) for _ in range(2))
fl = 2.0 * nq * ng * D
print(f"{nq} x {ng} x {D} cap {cap}: count {t:.3f} ms = {fl / t / 1e9:.1f} TF/s ({fl / t / 1e9 / 157.3:.3f} of f32 MFMA peak); plain distance matrix {td:.3f} ms ({fl / td / 1e9 / 157.3:.3f})")
