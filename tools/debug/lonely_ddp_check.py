"""Two ranks, one GPU, gloo: rank 0 trains on a batch with lonely identities (one real instance), rank 1 on a clean one.  The
device-mask step cannot raise inside the step; check_lonely_identities() must raise on BOTH ranks (the counter is all-reduced),
otherwise rank 1 would walk into its next collective alone.  Launched by tests/test_ddp_overlap_gpu.py under torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroids_reid_amd.bench_train import make_model, synthetic_batch      # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    torch.manual_seed(0)
    model = make_model(num_classes=40, dtype=torch.bfloat16, K=4)
    x, labels, cam, _ = synthetic_batch(8, 4, 64, 32, 0, rank, num_classes=40)
    real = np.ones(32, dtype=bool)
    if rank == 0:
        real[4:8] = [True, False, False, False]          # identity 1 keeps a single real instance
    model.training_step((x, labels, cam, torch.as_tensor(real).cuda()), 0)
    raised = False
    try:
        model.check_lonely_identities()
    except RuntimeError as e:
        raised = "1 real instance" in str(e)
    flag = torch.tensor([1 if raised else 0])
    dist.all_reduce(flag)                                   # would hang if one rank had left early
    print("LONELY_DDP_OK" if raised and int(flag) == 2 else f"LONELY_DDP_MISMATCH rank {rank} raised {raised} total {int(flag)}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
