"""Input side of the hot path: the PK sampler and the per-PID batch assembly whose
layout CTLModel.training_step relies on (PID-contiguous [P, K] batches with an `isReal` mask).

Mirrors datasets/samplers/distributed_pids_sampler.py:15-86 (RandomIdentitySampler) and
datasets/bases.py:339-406,447-455 (BaseDatasetLabelledPerPid.__getitem__, collate_fn_alternative): same
constructor arguments, same seeding (`random.seed(epoch)` per epoch), same consumption of the RNG streams, so
the emitted PID sequence and the per-PID instance choices are identical to the reference's for the same
inputs.  Host-side logic only (no GPU work); image decoding/augmentation stays out of scope -- `loader` is any
callable mapping the stored item to a CHW tensor.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch


class RandomIdentitySampler(torch.utils.data.Sampler):
    """Yields PID keys; every `batch_size` consecutive keys form one P-identity batch of this rank."""

    def __init__(self, data_source, batch_size, num_instances, world_size=None, rank=None):
        self.data_source = data_source
        self.batch_size = batch_size               # number of identities P per batch (reference naming)
        self.num_instances = num_instances
        self.num_pids_per_batch = batch_size
        self.world_size = world_size if world_size is not None else 1
        self.rank = rank if rank is not None else 0
        self.epoch = 0
        # how many times each identity can be visited in an epoch: ceil(n / K), an odd leftover image dropped
        self.visits = {}
        for pid, items in data_source.items():
            n = len(items)
            if n % num_instances == 1:
                n -= 1
            self.visits[pid] = int(math.ceil(n / num_instances))
        self.pids = list(self.visits.keys())
        self.length = sum(self.visits.values()) // self.world_size

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.length

    def __iter__(self):
        np.random.seed(self.epoch)
        random.seed(self.epoch)
        remaining = dict(self.visits)
        alive = list(self.pids)
        group = self.num_pids_per_batch * self.world_size
        order = []
        while len(alive) >= group:
            for pid in random.sample(alive, group):
                order.append(pid)
                remaining[pid] -= 1
                if remaining[pid] == 0:
                    alive.remove(pid)
        assert len(order) % group == 0
        mine = list(np.array_split(order, self.world_size)[self.rank])
        tail = len(mine) % self.batch_size
        if tail:
            mine = mine[:-tail]
        self.length = len(mine)
        return iter(mine)


class PerPidDataset(torch.utils.data.Dataset):
    """`samples[pid]` = list of (item, target, camid, idx); __getitem__(pid) -> K tuples
    (img, target, camid, idx, isReal) -- short identities are padded with zero images (isReal False) or
    resampled (isReal True), exactly like datasets/bases.py:346-406 (which also CONSUMES samples[pid])."""

    def __init__(self, data, loader, num_instances=4, resample=False):
        self.samples = data
        self.loader = loader
        self.num_instances = num_instances
        self.resample = resample

    def __len__(self):
        return len(self.samples) * self.num_instances

    def __getitem__(self, pid):
        pid = int(pid)
        K = self.num_instances
        snapshot = self.samples[pid][:]
        n = len(snapshot)
        assert n > 1, f"identity {pid} has {n} sample(s); at least 2 are required"
        take = min(n, K)
        random.shuffle(self.samples[pid])
        out = []
        for _ in range(take):
            item, target, camid, idx = self.samples[pid].pop(0)
            img = self.loader(item)
            out.append((img, target, camid, idx, True))
        if n < K:
            missing = K - n
            if self.resample:
                for j in np.random.choice(range(n), size=missing, replace=True):
                    item, target, camid, idx = snapshot[j]
                    img = self.loader(item)
                    out.append((img, target, camid, idx, True))
            else:
                blank = torch.zeros_like(img)
                out.extend((blank, target, camid, idx, False) for _ in range(missing))
        assert len(out) == K
        return out


def collate_pk(batch):
    """datasets/bases.py:447-455: flatten P lists of K tuples -> (x [P*K,...], pids int64, camids, isReal)."""
    flat = [t for sample in batch for t in sample]
    x = torch.stack([t[0] for t in flat], dim=0)
    pids = torch.tensor([t[1] for t in flat], dtype=torch.int64)
    return x, pids, torch.tensor([t[2] for t in flat]), torch.tensor([t[4] for t in flat])
