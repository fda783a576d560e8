"""Streamed evaluation's counting contraction (sqdist_count_f32_kernel) against its grid target CREID_STREAM_WGS (read once per
process: run this script once per value) on the three evaluation shapes of the bench -- DukeMTMC-shaped 2228 x 17661, the
north-star 3000 x 15000 and the per-rank shard of configs[3], 6250 x 200 000 (1.6 GB of gallery: far beyond the 256 MB Infinity
Cache, so how long the workgroups of one XCD stay in step on the same gallery tiles decides where the operands come from).
    for w in 512 1024 2048 4096 8192; do CREID_STREAM_WGS=$w python tools/debug/stream_wgs_probe.py; done"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import reid_metric as rm   # noqa: E402
from bench import time_kernel                      # noqa: E402

L = rm.L
lib = L.lib()
out = [f"CREID_STREAM_WGS={os.environ.get('CREID_STREAM_WGS', '512 (default)')} CREID_STREAM_TPER={os.environ.get('CREID_STREAM_TPER', '8 (default)')}"]
for nq, ng, npid in ((2228, 17661, 702), (3000, 15000, 702), (6250, 200_000, 50_000)):
    D = 2048
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
                                     L.ptr(plan.g_order), L.ptr(plan.q_cams), L.ptr(plan.g_cams), cap, L.ptr(pos_key),
                                     L.ptr(pos_idx), L.ptr(npos), L.stream()), "poslist")

    def count():
        L.check(lib.creid_stream_count(L.ptr(q), L.ptr(g), L.ptr(qq), L.ptr(gg), nq, ng, D, L.ptr(plan.q_pids), L.ptr(plan.g_pids),
                                       cap, L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), L.ptr(hist), L.stream()), "count")
    t = min(time_kernel(count, 3) for _ in range(2))
    tf = 2.0 * nq * ng * D / (t * 1e-3) / 1e12
    out.append(f"{nq}x{ng}: {t:.3f} ms {tf:.1f} TF/s ({tf / 157.3:.3f})")
    del feats, fn, q, g
    torch.cuda.empty_cache()
print("  ".join(out))
