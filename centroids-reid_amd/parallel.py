"""Data-parallel plumbing (one process per GPU, torch.distributed: RCCL over xGMI on the GPUs, gloo on
CPU for tests).  The reference trains with PL's DDP (utils/misc.py:101-119: gradients all-reduced,
everything else rank-local, no SyncBN) and evaluates on rank 0 only (modelling/bases.py:169,300).
Here:
  * training: ONE collective per step -- all-reduce(mean) of the flat fp32 gradient buffer (all Adam
    parameters are views into it) plus the center-loss gradient; BN statistics, mining and centroids stay
    rank-local exactly like the reference;
  * evaluation: embeddings are all-gathered so every rank holds the full gallery, QUERY ROWS are sharded
    across ranks (distance + rank + CMC/AP locally), and only the per-query results (valid, AP, first-match
    rank) are gathered -- tiny -- before the means of utils/eval_reid.py:86-90 are taken.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

K_LIST = (1, 5, 10, 20, 50)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_mean_(buffers, world=None):
    """In-place mean over ranks of each tensor in `buffers` (one collective per tensor)."""
    rank, w = world_info()
    world = world or w
    if world == 1:
        return
    for b in buffers:
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        b.mul_(1.0 / world)


def make_grad_sync(world):
    """grad_sync hook for CTLModel: all-reduce the flat Adam gradient buffer and the centers' gradient;
    the 1/world of the big buffer is folded into the Adam kernel (grad_scale) instead of a separate pass."""
    def sync(model):
        opt, _ = model.optimizers()
        dist.all_reduce(opt.gflat, op=dist.ReduceOp.SUM)
        opt.grad_scale = 1.0 / world
        cg = model.center_loss.centers.grad
        dist.all_reduce(cg, op=dist.ReduceOp.SUM)
        cg.mul_(1.0 / world)
    return sync


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank` (np.array_split sizes, like the reference's PK sampler
    split, datasets/samplers/distributed_pids_sampler.py:71)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, counts=None) -> torch.Tensor:
    """Concatenate row-shards [n_r, D] from all ranks (ragged allowed)."""
    rank, world = world_info()
    if world == 1:
        return local
    if counts is None:
        cnt = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        counts = [int(c.item()) for c in allc]
    mx = max(counts)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def merge_eval_results(valid, ap, first, max_rank=50):
    """utils/eval_reid.py:86-90 over per-query results: (cmc float32[max_rank], mAP, topk float64[5])."""
    valid = np.asarray(valid, bool); ap = np.asarray(ap, np.float64); first = np.asarray(first, np.int64)
    nv = float(valid.sum())
    f = first[valid]
    cmc = ((f[:, None] <= np.arange(max_rank)[None, :]).astype(np.float32).sum(0) / nv).astype(np.float32)
    mAP = float(np.mean(ap[valid]))
    topk = np.stack([(f < k) for k in K_LIST], axis=1).astype(np.int64).mean(axis=0)
    return cmc, mAP, topk


def evaluate_sharded(feats_local, pids, camids, num_query, per_query_fn, feat_counts=None, max_rank=50):
    """Distributed R1_mAP: `feats_local` is this rank's row-shard of the [nq + ng, D] embeddings (in global
    order); `per_query_fn(feats_full, q_lo, q_hi) -> (valid, ap, first)` ranks query rows [q_lo, q_hi)
    against the whole gallery (the HIP kernels on a GPU; any callable in tests)."""
    rank, world = world_info()
    feats = all_gather_rows(feats_local, feat_counts)
    lo, hi = shard_bounds(num_query, rank, world)
    v, a, f = per_query_fn(feats, lo, hi)
    pack = torch.stack([torch.as_tensor(np.asarray(v), dtype=torch.float64),
                        torch.as_tensor(np.asarray(a), dtype=torch.float64),
                        torch.as_tensor(np.asarray(f), dtype=torch.float64)], dim=1).to(feats_local.device)
    counts = [shard_bounds(num_query, r, world)[1] - shard_bounds(num_query, r, world)[0] for r in range(world)]
    allp = all_gather_rows(pack, counts).cpu().numpy()
    return merge_eval_results(allp[:, 0] > 0.5, allp[:, 1], allp[:, 2].astype(np.int64), max_rank)
