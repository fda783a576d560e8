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
        opt, opt_c = model.optimizers()
        opt, opt_c = getattr(opt, "_optimizer", opt), getattr(opt_c, "_optimizer", opt_c)
        dist.all_reduce(opt.gflat, op=dist.ReduceOp.SUM)          # (the centers' gradient rides in its tail: solver.build_optimizer)
        opt.grad_scale = 1.0 / world
        if getattr(opt_c, "grad_in_adam_tail", False):          # (re-checked per call: .grad may have been rebound, solver.CenterSGD)
            opt_c.grad_scale = 1.0 / world
        else:
            if hasattr(opt_c, "grad_scale"):
                opt_c.grad_scale = 1.0
            cg = model.center_loss.centers.grad
            dist.all_reduce(cg, op=dist.ReduceOp.SUM)
            cg.mul_(1.0 / world)
    return sync


class GradBuckets:
    """Reverse-order contiguous buckets of the flat fp32 gradient buffer for overlap with backward.

    The flat buffer holds the parameters in `named_parameters()` order (stem, layer1..layer4, BNNeck, classifier), and
    backward produces gradients back to front, so the ranges
        bucket 0 = [layer4 .. end of buffer]   (layer4 + heads: final when layer4's backward is done)
        bucket 1 = [layer3 .. layer4)
        bucket 2 = [start  .. layer3)          (stem, layer1, layer2)
    become final in that order and each is ONE contiguous slice: a bucket is all-reduced on a side stream as soon as
    its last kernel has been enqueued, while the next layer group's backward runs (PL's DDP does the same with its
    reverse-order 25 MB buckets, utils/misc.py:101-119; here the sizes follow the layer groups: ~66 / 28 / 6 MB for
    ResNet50, few large transfers for the per-link-bound xGMI rings).  The 1/world is folded into the Adam kernel."""

    def __init__(self, names, offsets, total, groups=("layer4.", "layer3.")):
        """names / offsets: parameter names and start offsets (elements) in flat-buffer order; total: padded length."""
        cuts = []
        for gname in groups:
            start = next((o for n, o in zip(names, offsets) if gname in n), None)
            if start is not None and start not in cuts and 0 < start < total:
                cuts.append(start)
        cuts = sorted(set(cuts), reverse=True)
        hi = total
        self.ranges = []
        for c in cuts:
            self.ranges.append((c, hi)); hi = c
        self.ranges.append((0, hi))
        self.total = total

    @classmethod
    def for_optimizer(cls, opt, groups=("layer4.", "layer3.")):
        names = opt.param_groups[0].get("names") or [str(i) for i in range(len(opt._params))]
        return cls(list(names), list(opt._offsets), opt.gflat.numel(), groups)

    def __len__(self):
        return len(self.ranges)

    def all_reduce(self, gflat, i):
        lo, hi = self.ranges[i]
        if hi > lo:
            dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM)


def _layer_number(group_name: str) -> int:
    """"layer4." -> 4 (the backbone engine's group hook is called with the layer number)."""
    import re
    m = re.fullmatch(r"layer(\d+)\.?", group_name)
    if m is None:
        raise ValueError(f"gradient bucket groups are named 'layer<k>.': got {group_name!r}")
    return int(m.group(1))


def _default_groups():
    """Layer groups whose gradient buckets are all-reduced as they become final.  Default: layer4 (+ heads + the centers'
    gradient in the buffer's tail) | layer3 | the rest = 3 collectives per step; CREID_DDP_GROUPS="layer4." makes it 2
    (the last one then carries 34 MB instead of 6 MB with nothing left to hide it behind)."""
    import os
    e = os.environ.get("CREID_DDP_GROUPS")
    return tuple(g for g in e.split(",") if g) if e else ("layer4.", "layer3.")


def make_overlapped_grad_sync(model, world, groups=None):
    """Returns (buckets, on_group_done, finish): `on_group_done(k)` is the backbone engine's backward hook (k = 4, 3,
    2, 1 after layer k's last kernel has been enqueued): it all-reduces the bucket that just became final on a side
    stream; `finish()` reduces what is left (first bucket, centers) and makes the main stream wait for the side
    stream before the optimiser kernels.  Works with eager steps and with the step captured as per-bucket graph
    segments (bench_train.DDPStepper)."""
    groups = _default_groups() if groups is None else groups
    opt, opt_c = model.optimizers()
    opt, opt_c = getattr(opt, "_optimizer", opt), getattr(opt_c, "_optimizer", opt_c)
    buckets = GradBuckets.for_optimizer(opt, groups)              # (the first bucket ends at the buffer's end: tail included)
    opt.grad_scale = 1.0 / world
    if hasattr(opt_c, "readopt_tail"):
        opt_c.readopt_tail()                                      # gradients rebound before the sync was built: back into the tail
    if hasattr(opt_c, "grad_scale"):
        opt_c.grad_scale = 1.0 / world if getattr(opt_c, "grad_in_adam_tail", False) else 1.0
    side = torch.cuda.Stream()
    layer_to_bucket = {}
    for i, g in enumerate(groups):
        if i < len(buckets) - 1:
            layer_to_bucket[_layer_number(g)] = i
    state = {"next": 0}

    def _launch(i):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buckets.all_reduce(opt.gflat, i)

    def on_group_done(k):
        i = layer_to_bucket.get(k)
        if i is not None and i == state["next"]:
            _launch(i)
            state["next"] = i + 1

    def finish():
        while state["next"] < len(buckets):
            _launch(state["next"]); state["next"] += 1
        # evaluated at every step: a gradient rebound away from the tail since the last one takes the explicit all-reduce
        tail = getattr(opt_c, "grad_in_adam_tail", False)
        if hasattr(opt_c, "grad_scale"):
            opt_c.grad_scale = 1.0 / world if tail else 1.0
        if not tail:
            cg = model.center_loss.centers.grad
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_reduce(cg, op=dist.ReduceOp.SUM)
                cg.mul_(1.0 / world)
        torch.cuda.current_stream().wait_stream(side)
        state["next"] = 0

    return buckets, on_group_done, finish


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank` (np.array_split sizes, like the reference's PK sampler
    split, datasets/samplers/distributed_pids_sampler.py:71)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, counts=None, force_collective=False) -> torch.Tensor:
    """Concatenate row-shards [n_r, D] from all ranks (ragged allowed) with ONE all_gather_into_tensor: equal shards
    land directly in the output; ragged shards are padded to the longest and the padding rows are dropped after.
    force_collective: issue the collective even in a one-rank group (first-contact test of the RCCL path)."""
    rank, world = world_info()
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local
    if counts is None:
        cnt = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        allc = torch.empty(world, device=local.device, dtype=torch.int64)
        dist.all_gather_into_tensor(allc, cnt)
        counts = [int(c) for c in allc.tolist()]
    mx = max(counts)
    tail = tuple(local.shape[1:])
    pad = local.contiguous()
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tail)])
    out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    if all(c == mx for c in counts):
        return out
    out = out.view((world, mx) + tail)
    return torch.cat([out[r, :c] for r, c in enumerate(counts)])


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
