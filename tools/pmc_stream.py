"""Workload for a rocprofv3 --pmc FETCH_SIZE pass over the streamed evaluation's counting contraction on the per-rank shard of
configs[3] (6250 x 200 000 x 2048): ONE warm launch + ONE counted launch of sqdist_count_f32_kernel.  Run once per grid rule:
    CREID_STREAM_TPER=1000 rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python tools/pmc_stream.py     (131 tiles per workgroup)
    CREID_STREAM_TPER=8    ...                                                                               (the shipped rule)
FETCH_SIZE counts 0.5 x the bytes of 16-byte streaming loads on this part (profiles/r0x_pmc_traffic.json calibration)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centroids_reid_amd import reid_metric as rm   # noqa: E402

L = rm.L
lib = L.lib()
nq, ng, D, npid = 6250, 200_000, 2048, 50_000
gen = torch.Generator(device="cuda").manual_seed(4)
feats = torch.randn((nq + ng, D), generator=gen, device="cuda", dtype=torch.float32)
rng = np.random.default_rng(4)
pids = np.concatenate([rng.integers(0, npid, nq), np.arange(ng) % npid])
cams = np.concatenate([np.zeros(nq, np.int64), np.ones(ng, np.int64)])
plan = rm.StreamPlan.on_device(pids, cams, nq, "cuda").finish()
fn, sq = rm.l2_normalize(feats, return_sqnorm=True)
q, g = fn[:nq], fn[nq:]
qq, gg = sq[:nq].contiguous(), sq[nq:].contiguous()
cap = plan.cap
pos_key = torch.empty((nq, cap), dtype=torch.int32, device="cuda"); pos_idx = torch.empty_like(pos_key)
npos = torch.empty(nq, dtype=torch.int32, device="cuda"); hist = torch.zeros((nq, cap), dtype=torch.int32, device="cuda")
L.check(lib.creid_stream_poslist(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_slot), L.ptr(plan.csr_off),
                                 L.ptr(plan.g_order), L.ptr(plan.q_cams), L.ptr(plan.g_cams), cap, L.ptr(pos_key), L.ptr(pos_idx),
                                 L.ptr(npos), L.stream()), "poslist")
for _ in range(2):
    L.check(lib.creid_stream_count(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_pids), L.ptr(plan.g_pids), cap,
                                   L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), L.ptr(hist), L.stream()), "count")
torch.cuda.synchronize()
print("PMCMETA", {"algorithmic_bytes": (nq + ng) * D * 4, "tper": os.environ.get("CREID_STREAM_TPER", "8")})
