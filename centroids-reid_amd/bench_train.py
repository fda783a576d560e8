"""Training-step benchmark / smoke helpers (BASELINE configs[1]: ResNet50 256x128 bf16, P=16 x K=4)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .config import get_cfg_defaults
from .train_ctl_model import CTLModel

R50_FWD_BWD_GFLOP_PER_IMG = 24.32     # BASELINE.md section 3 (3 x forward conv FLOPs)
MFMA_BF16_TFLOPS = 2500.0


def make_model(num_classes=751, dtype=torch.bfloat16, arch="resnet50", K=4):
    cfg = get_cfg_defaults()
    cfg.MODEL.PRETRAINED = False
    cfg.MODEL.NAME = arch
    cfg.DATALOADER.NUM_INSTANCE = K
    cfg.USE_MIXED_PRECISION = dtype != torch.float32
    model = CTLModel(cfg, num_classes=num_classes, num_query=0, compute_dtype=dtype).cuda().train()
    model.configure_optimizers()
    return model


def synthetic_batch(P, K, H, W, step, rank=0, num_classes=751, seed=0):
    """Synthetic PK batch on the device: N(0,1) images, pid = (arange(P)*7 + step*P*world...) mod C."""
    gen = torch.Generator(device="cuda").manual_seed(seed * 1000 + rank * 100 + step)
    x = torch.randn((P * K, 3, H, W), generator=gen, device="cuda", dtype=torch.float32)
    pids = (np.arange(P) * 7 + step * 13 + rank * 97) % num_classes
    labels = torch.as_tensor(np.repeat(pids, K).astype(np.int64), device="cuda")
    camid = torch.zeros(P * K, dtype=torch.int64)
    is_real = torch.ones(P * K, dtype=torch.bool)          # stays on the host: no per-step D2H sync
    return x, labels, camid, is_real


def make_grad_sync(world):
    def sync(model):
        opt, opt_center = model.optimizers()
        dist.all_reduce(opt.gflat)
        opt.grad_scale = 1.0 / world
        cg = model.center_loss.centers.grad
        dist.all_reduce(cg)
        cg.mul_(1.0 / world)
    return sync


def run(args, rank, world, barrier_sync, time_kernel):
    P, K, H, W = 16, 4, 256, 128
    model = make_model()
    if world > 1:
        # identical initial weights on every rank, then data-parallel gradient all-reduce over RCCL
        opt, _ = model.optimizers()
        dist.broadcast(opt.flat, 0)
        dist.broadcast(model.center_loss.centers.data, 0)
        for b in model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, 0)
        model.grad_sync = make_grad_sync(world)
    batches = [synthetic_batch(P, K, H, W, s, rank) for s in range(4)]
    use_graph = os.environ.get("CREID_NO_GRAPH", "0") != "1" and world == 1
    if use_graph:
        # The whole step (prep, fwd, losses, bwd, optimiser kernels) is captured ONCE into a hipGraph; every
        # timed step copies a fresh synthetic batch into the static input buffers and replays it.
        sx, sl = batches[0][0].clone(), batches[0][1].clone()
        static = (sx, sl, batches[0][2], batches[0][3])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for s in range(3):
                model.training_step(static, s)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gout = model.training_step(static, 0)

        def one_step(s):
            sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
            graph.replay()
            return gout
    else:
        def one_step(s):
            return model.training_step(batches[s % 4], s)
    for s in range(args.warmup):
        out = one_step(s)
    barrier_sync(world)
    t0 = time.perf_counter()
    for s in range(args.steps):
        out = one_step(s)
    t_host = time.perf_counter() - t0          # host enqueue time (diagnostic: host- vs GPU-bound)
    barrier_sync(world)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    imgs = P * K * world * args.steps
    res = {}
    if rank == 0:
        loss = float(out["loss"])
        assert np.isfinite(loss), "non-finite loss in the benchmark"
        res["final_loss"] = loss
        res["host_enqueue_ms_per_step"] = t_host / args.steps * 1e3
        res["hip_graph"] = bool(use_graph)
        ms = dt / args.steps * 1e3
        tf = R50_FWD_BWD_GFLOP_PER_IMG * P * K / (ms * 1e-3) / 1e3
        res["roofline"] = {"kernel": "whole step (conv fwd+dgrad+wgrad MFMA work / step time)", "bound": "mfma",
                           "achieved": tf, "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_TFLOPS,
                           "traffic": None}
    return {"metric": "train_images_per_sec", "value": imgs / dt, "unit": "images/s",
            "ms_per_step": dt / args.steps * 1e3, "dtype": "bf16",
            "config": {"workload": "ResNet50 256x128 CTL training step: fwd+bwd, centroid-triplet + center + xent, "
                                   "Adam + center SGD (BASELINE configs[1])",
                       "P": P, "K": K, "global_batch": P * K * world, "num_classes": 751,
                       "parallelism": f"dp{world}" if world > 1 else "single"}, **res}
