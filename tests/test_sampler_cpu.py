"""CPU: PK sampler + per-PID batch assembly against sequences recorded from the reference
(tools/gen_golden.py sampler).  Integer/index work: bit-exact."""
import json
import random

import numpy as np
import torch


def _table(g):
    raw = json.loads(str(g["table"]))
    return {int(k): [tuple(t) for t in v] for k, v in raw.items()}


def test_random_identity_sampler_sequences(golden):
    from centroids_reid_amd.sampler import RandomIdentitySampler
    g = golden("sampler")
    table = _table(g)
    src = {p: [t[3] for t in v] for p, v in table.items()}
    for world in (1, 2):
        for rank in range(world):
            s = RandomIdentitySampler(src, 4, 4, world, rank)
            for ep in (0, 1):
                s.set_epoch(ep)
                seq = np.asarray([int(x) for x in s], np.int64)
                np.testing.assert_array_equal(seq, g[f"w{world}_r{rank}_e{ep}"])
                assert len(s) == int(g[f"w{world}_r{rank}_e{ep}_len"])
    # (reference quirk kept: np.array_split cuts the EPOCH sequence into contiguous per-rank halves, so the
    #  i-th batches of two ranks are not the halves of one P*world group and may share identities)


def test_per_pid_dataset_and_collate(golden):
    from centroids_reid_amd.sampler import PerPidDataset, collate_pk
    g = golden("sampler")
    loader = lambda path: torch.full((1, 2, 2), float(int(path[3:].split("_")[0]) * 100 + int(path.split("_")[1])))
    for resample in (False, True):
        ds = PerPidDataset({p: list(v) for p, v in _table(g).items()}, loader, 4, resample)
        random.seed(5); np.random.seed(5)
        rows, batch = [], []
        for pid in (0, 3, 7, 3, 11, 20):
            if len(ds.samples[pid]) <= 1:
                continue
            out = ds[pid]
            batch.append(out)
            rows.append([[float(t[0].flatten()[0]), t[1], t[2], t[3], int(t[4])] for t in out])
        np.testing.assert_array_equal(np.asarray(rows, np.float64), g[f"items_resample{int(resample)}"])
        x, pids, cams, is_real = collate_pk(batch)
        assert x.shape[0] == 4 * len(batch) and pids.dtype == torch.int64 and is_real.dtype == torch.bool
        assert (pids.view(len(batch), 4) == pids.view(len(batch), 4)[:, :1]).all()      # PID-contiguous [P, K]
        if not resample:
            assert (x[~is_real] == 0).all()
