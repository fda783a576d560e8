"""Control-flow check of the overlapped data-parallel step on ONE GPU: two processes share cuda:0 over gloo.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/debug/ddp_overlap_check.py
Checks: segmented-graph + bucketed side-stream all-reduce == eager flat all-reduce (same weights after 3 steps, both
ranks identical), graph segments = 3 + optimiser."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ".")
from centroids_reid_amd import bench_train as bt, parallel, ops  # noqa: E402

ops._DETERMINISTIC = True
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
P, K, H, W = 4, 4, 128, 64


def fresh():
    torch.manual_seed(0)
    m = bt.make_model(num_classes=60, K=K)
    opt, _ = m.optimizers()
    dist.broadcast(opt.flat, 0); m.backbone.engine.weights_dirty = True
    dist.broadcast(m.center_loss.centers.data, 0)
    return m, opt


batches = [bt.synthetic_batch(P, K, H, W, s, rank, num_classes=60) for s in range(3)]
res = {}
for mode in ("flat_eager", "overlap_eager", "overlap_graph"):
    m, opt = fresh()
    sx, sl = batches[0][0].clone(), batches[0][1].clone()
    static = (sx, sl, batches[0][2], batches[0][3])
    if mode == "flat_eager":
        m.grad_sync = parallel.make_grad_sync(world)
        step = lambda s: m.training_step(static, s)
    else:
        snap = (opt.flat.clone(), m.center_loss.centers.detach().clone(), {k: v.clone() for k, v in m.state_dict().items()})
        st = bt.DDPStepper(m, world, static, use_graph=(mode == "overlap_graph"))
        if mode == "overlap_graph":
            assert len(st.segs) == 3 and st.split_at == [4, 3], (len(st.segs), st.split_at)
            with torch.no_grad():                      # the stepper's warm-up advanced the state: rewind
                m.load_state_dict(snap[2]); opt.flat.copy_(snap[0]); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.hyper[1:].zero_()
        step = st.step
    losses = []
    for s in range(3):
        sx.copy_(batches[s][0]); sl.copy_(batches[s][1])
        losses.append(float(step(s)["loss"]))
    torch.cuda.synchronize()
    w = opt.flat.detach().clone()
    other = w.clone()
    dist.broadcast(other, 0)
    res[mode] = (losses, w.cpu(), bool(torch.equal(other, w)))
    dist.barrier()
ok = True
for mode, (losses, w, same) in res.items():
    print(rank, mode, losses, "ranks identical:", same, flush=True)
    ok &= same
ok &= bool(torch.equal(res["flat_eager"][1], res["overlap_eager"][1]))
ok &= bool(torch.equal(res["flat_eager"][1], res["overlap_graph"][1]))
print(rank, "DDP_OVERLAP_OK" if ok else "DDP_OVERLAP_MISMATCH", flush=True)
dist.destroy_process_group()
