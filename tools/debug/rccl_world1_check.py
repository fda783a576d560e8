"""First contact with RCCL on a one-GPU box: backend "nccl" (= RCCL on ROCm) with a ONE-rank group.
    python tools/debug/rccl_world1_check.py
Exercises what the 8-GPU run will use -- communicator init on the device, all-reduce of the flat gradient buffer on the
step's stream, the bucketed all-reduces on a SIDE stream between hipGraph segments (bench_train.DDPStepper),
all_gather_into_tensor of embeddings, the query-sharded evaluation -- and checks every schedule against the step without any
process group: identical weights after 3 steps, bit for bit (a one-rank sum is the identity and grad_scale is 1)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroids_reid_amd import bench_train as bt, parallel, ops, reid_metric as rm  # noqa: E402

ops._DETERMINISTIC = True
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
P, K, H, W = 4, 4, 128, 64
batches = [bt.synthetic_batch(P, K, H, W, s, 0, num_classes=60) for s in range(3)]


def fresh():
    torch.manual_seed(0)
    m = bt.make_model(num_classes=60, K=K)
    opt, _ = m.optimizers()
    return m, opt


res = {}
for mode in ("no_group", "flat_rccl", "overlap_eager_rccl", "overlap_graph_rccl"):
    m, opt = fresh()
    sx, sl = batches[0][0].clone(), batches[0][1].clone()
    static = (sx, sl, batches[0][2], batches[0][3])
    if mode == "no_group":
        step = lambda s: m.training_step(static, s)
    elif mode == "flat_rccl":
        m.grad_sync = parallel.make_grad_sync(1)
        step = lambda s: m.training_step(static, s)
    else:
        snap = (opt.flat.clone(), {k: v.clone() for k, v in m.state_dict().items()})
        st = bt.DDPStepper(m, 1, static, use_graph=(mode == "overlap_graph_rccl"))
        if mode == "overlap_graph_rccl":
            assert len(st.segs) == 3 and st.split_at == [4, 3], (len(st.segs), st.split_at)
            with torch.no_grad():                      # the stepper's warm-up advanced the state: rewind
                m.load_state_dict(snap[1]); opt.flat.copy_(snap[0]); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.hyper[1:].zero_()
        step = st.step
    losses = []
    for s in range(3):
        sx.copy_(batches[s][0]); sl.copy_(batches[s][1])
        losses.append(float(step(s)["loss"]))
    torch.cuda.synchronize()
    res[mode] = (losses, opt.flat.detach().clone().cpu(), m.center_loss.centers.detach().clone().cpu())
    print(mode, losses, flush=True)
ok = True
for mode in res:
    same = bool(torch.equal(res[mode][1], res["no_group"][1])) and bool(torch.equal(res[mode][2], res["no_group"][2]))
    print(mode, "== no_group:", same, flush=True)
    ok &= same and res[mode][0] == res["no_group"][0]

# ---- evaluation side: all_gather_into_tensor of embeddings + query-sharded streamed evaluation
rng = np.random.default_rng(0)
nq, ng, D = 96, 1500, 256
feats = torch.from_numpy(rng.standard_normal((nq + ng, D)).astype(np.float32)).cuda()
pids = rng.integers(0, 50, nq + ng); cams = rng.integers(0, 4, nq + ng)
g = parallel.all_gather_rows(feats, [nq + ng], force_collective=True)
ok &= bool(torch.equal(g, feats))
ref = rm.R1_mAP(num_query=nq, streamed=True)
cmc0, map0, topk0 = ref.compute(feats, pids, cams)


def per_query(full, lo, hi):
    r = rm.R1_mAP(num_query=hi - lo, streamed=True)
    sub = torch.cat([full[lo:hi], full[nq:]])
    r.compute(sub, np.concatenate([pids[lo:hi], pids[nq:]]), np.concatenate([cams[lo:hi], cams[nq:]]))
    return r.last["valid"].cpu().numpy() == 1, r.last["ap"].cpu().numpy(), r.last["first"].cpu().numpy()


cmc1, map1, topk1 = parallel.evaluate_sharded(feats, pids, cams, nq, per_query, feat_counts=[nq + ng])
ok &= abs(map1 - map0) < 1e-12 and np.array_equal(cmc1, cmc0) and np.allclose(topk1, topk0, atol=1e-12)
print("eval sharded == single:", abs(map1 - map0) < 1e-12, flush=True)
print("RCCL_WORLD1_OK" if ok else "RCCL_WORLD1_MISMATCH", flush=True)
dist.destroy_process_group()
