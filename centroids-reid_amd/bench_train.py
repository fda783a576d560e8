"""Training-step benchmark / smoke helpers (BASELINE configs[1]: ResNet50 256x128 bf16, P=16 x K=4)."""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .config import get_cfg_defaults
from .train_ctl_model import CTLModel
from . import parallel

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


def conv_shapes(B, H, W, last_stride=1):
    """(cin, cout, k, stride, Hin, Win) of every non-stem convolution of ResNet50 at this input size."""
    shapes, h, w, inpl = [], H // 4, W // 4, 64
    for planes, n, st in zip((64, 128, 256, 512), (3, 4, 6, 3), (1, 2, 2, last_stride)):
        for b in range(n):
            s = st if b == 0 else 1
            shapes.append((inpl, planes, 1, 1, h, w))
            shapes.append((planes, planes, 3, s, h, w))
            ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            shapes.append((planes, planes * 4, 1, 1, ho, wo))
            if b == 0:
                shapes.append((inpl, planes * 4, 1, s, h, w))
            inpl, h, w = planes * 4, ho, wo
    return shapes


def igemm_roofline(B, H, W, time_kernel, reps=5):
    """Live HIP-event timing of the dominant kernel family (igemm_bf16_kernel: conv forward + data gradient)
    over the real ResNet50 layer mix: achieved = sum(algorithmic FLOPs) / sum(avg launch duration)."""
    from . import layers as ly
    tot_flop = tot_ms = 0.0
    worst = []
    for cin, cout, k, s, h, w in conv_shapes(B, H, W):
        x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
        wt = (torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5)
        krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
        pad = k // 2
        ms_f = time_kernel(lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), reps)
        y = ly.conv2d_fwd(x, krsc, s, pad)
        ms_d = time_kernel(lambda: ly.conv2d_dgrad(y, crsk, (h, w), s, pad), reps)
        fl = 2.0 * B * y.shape[1] * y.shape[2] * cout * cin * k * k
        tot_flop += 2 * fl; tot_ms += ms_f + ms_d
        worst.append((fl / (ms_f * 1e-3) / 1e12, f"{cin}->{cout} k{k} s{s} {h}x{w}"))
    worst.sort()
    return tot_flop / (tot_ms * 1e-3) / 1e12, tot_ms, worst[:3], worst[-3:]


def hbm_stage_rates(time_kernel, B, H, W):
    """Achieved GB/s of the bandwidth-bound passes of the step on their largest instance (layer1 block output,
    [B*H/4*W/4, 256] bf16): BN apply (+residual, ReLU), BN backward (reduce + finalize + apply), and Adam over the
    flat parameter buffer -- the HBM side of the roofline report (peak 8 TB/s spec, ~6.3 TB/s achievable)."""
    from . import layers as ly
    from . import _lib as L
    M, Cc = B * (H // 4) * (W // 4), 256
    dev = "cuda"
    x = torch.randn((M, Cc), device=dev).to(torch.bfloat16)
    res = torch.randn((M, Cc), device=dev).to(torch.bfloat16)
    g = torch.randn((M, Cc), device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    y, mean, invstd = ly.bn2d_train_fwd(x, gamma, beta, rm, rv, residual=res, relu=True)
    ss = torch.stack([gamma * invstd, beta - mean * gamma * invstd]).contiguous()
    out = torch.empty_like(x)
    lib = L.lib()
    t_apply = time_kernel(lambda: L.check(lib.creid_bn2d_apply(L.ptr(x), L.ptr(ss), L.ptr(res), 1, M, Cc, 1, L.ptr(out),
                                                                  L.stream()), "bn2d_apply"), 20)
    t_bwd = time_kernel(lambda: ly.bn2d_bwd(x, g, y, mean, invstd, gamma, want_gm=True), 20)
    n = 25_600_000
    p, gr, m1, v1 = (torch.zeros(n, device=dev) for _ in range(4))
    hyper = torch.tensor([3.5e-4, 1.0, 0.1, 0.03], device=dev)
    t_adam = time_kernel(lambda: L.check(lib.creid_adam_step_dev(L.ptr(p), L.ptr(gr), L.ptr(m1), L.ptr(v1), n, L.ptr(hyper),
                                                                    0.9, 0.999, 1e-8, 5e-4, 1.0, L.stream()), "adam"), 10)
    e = M * Cc * 2
    return {"bn2d_apply_GBs": 3 * e / (t_apply * 1e-3) / 1e9,                 # read x, residual; write y
            "bn2d_bwd_GBs": (3 * e + 3 * e + 2 * e) / (t_bwd * 1e-3) / 1e9,     # reduce: x,g,act; apply: x,g,act -> dx,gm
            "adam_GBs": 7 * n * 4 / (t_adam * 1e-3) / 1e9,                     # read p,g,m,v; write p,m,v
            "peak_GBs": 8000.0}


def pmc_field(key, field):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        with open(os.path.join(root, "profiles", "r02_pmc_traffic.json")) as f:
            return json.load(f)[key][field]
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(key):
    """Average HBM-side bytes per launch of a kernel family from the committed rocprofv3 --pmc passes
    (profiles/r02_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE -- FETCH_SIZE reads 0.5x on 16-byte streaming loads, see
    the calibration block of that file -- collected in their own counter-only runs); None if absent."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(root, "profiles", name)) as f:
                e = json.load(f)[key]
            return (e["fetch_bytes"] + e["write_bytes"]) / e["launches"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def map_delta_bf16(n_id=96, n_query=2, n_gallery=6, H=256, W=128, noise=0.6, seed=0):
    """BASELINE metric (iii) on synthetic data: mAP of the retrieval evaluation on embeddings of CLUSTERED synthetic
    identities (image = smooth per-identity pattern + N(0, noise) pixels), once through the fp32 parity mode and once
    through the bf16 throughput mode of the same randomly initialised ResNet50 (BatchNorm running statistics settled
    by a few training-mode passes first, then eval-mode embedding + BNNeck like validation_step).  Random weights are
    a random feature extractor: identities stay separable, camera / instance noise does the rest."""
    from . import reid_metric as rm
    torch.manual_seed(seed)                                    # the random initialisation is part of the recipe
    gen = torch.Generator(device="cuda").manual_seed(seed)
    per = n_query + n_gallery
    base = torch.randn((n_id, 3, H // 16, W // 16), generator=gen, device="cuda")
    base = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    x = base.repeat_interleave(per, 0) + noise * torch.randn((n_id * per, 3, H, W), generator=gen, device="cuda")
    pid = np.repeat(np.arange(n_id), per)
    slot = np.tile(np.arange(per), n_id)
    q_rows = np.nonzero(slot < n_query)[0]; g_rows = np.nonzero(slot >= n_query)[0]
    order = np.concatenate([q_rows, g_rows])
    pids = pid[order]
    cams = np.concatenate([np.zeros(len(q_rows), np.int64), np.ones(len(g_rows), np.int64)])   # datasets/bases.py:226-229
    ref = make_model(dtype=torch.float32)
    ref.train()
    with torch.no_grad():
        for s in range(0, min(len(x), 512), 64):                           # settle the running statistics ONCE (fp32)
            _, f = ref.backbone(x[s:s + 64])
            ref.bn(f)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    del ref
    embs = {}
    for dt in (torch.float32, torch.bfloat16):
        model = make_model(dtype=dt)
        model.load_state_dict(sd)                                          # identical weights AND statistics
        model.eval()
        out = []
        with torch.no_grad():
            for s in range(0, len(x), 64):
                _, f = model.backbone(x[s:s + 64])
                out.append(model.bn(f).float())
        embs[dt] = torch.cat(out)[torch.as_tensor(order, device="cuda")].contiguous()
        del model
    res = {}
    for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        cmc, mAP, _ = rm.R1_mAP(num_query=len(q_rows), streamed=True).compute(embs[dt], pids, cams)
        res[name] = (mAP, float(cmc[0]))
    a, b = embs[torch.float32], embs[torch.bfloat16]
    cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
    return {"mAP_f32": res["f32"][0], "mAP_bf16": res["bf16"][0], "mAP_bf16_minus_f32": res["bf16"][0] - res["f32"][0],
            "rank1_f32": res["f32"][1], "rank1_bf16": res["bf16"][1], "min_cosine": float(cos.min().item()),
            "images": int(len(x)), "identities": n_id, "noise": noise}


class DDPStepper:
    """Data-parallel training step with the gradient all-reduce OVERLAPPED with backward and no eager kernels between
    the captured pieces: the step is captured as hipGraph segments split where a gradient bucket becomes final
    (after layer4's and layer3's backward), each bucket's RCCL all-reduce is enqueued on a side stream right after
    its segment's replay and runs under the next segment, and the optimiser segment waits for the side stream.
        seg0 = forward + heads + layer4 backward   -> all-reduce(layer4 + heads, ~66 MB)   [side stream]
        seg1 = layer3 backward                     -> all-reduce(layer3, ~28 MB)           [side stream]
        seg2 = layer2, layer1, stem backward       -> all-reduce(rest, ~6 MB) + centers    [side stream]
        seg3 = Adam + center SGD (after the side stream)
    The collectives themselves stay outside the graphs (RCCL calls are issued by the host on the side stream)."""

    def __init__(self, model, world, static, use_graph=True):
        self.model, self.world, self.static = model, world, static
        self.buckets, self.hook, self.finish = parallel.make_overlapped_grad_sync(model, world)
        self.eng = model.backbone.engine
        self.use_graph = use_graph
        self.segs, self.split_at, self.gopt, self.out = [], [], None, None
        model.grad_sync = None
        if use_graph:
            self._capture()

    def _capture(self):
        model, eng = self.model, self.eng
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for s in range(2):                              # eager warm-up (allocations, lazy inits), with real syncs
                self._eager_step(s)
            torch.cuda.synchronize()
            segs = [torch.cuda.CUDAGraph()]
            split_at = []

            def cap_hook(k):
                if k in (4, 3):
                    segs[-1].capture_end()
                    split_at.append(k)
                    g = torch.cuda.CUDAGraph()
                    segs.append(g)
                    g.capture_begin(pool=segs[0].pool())
            eng.on_group_done = cap_hook
            segs[0].capture_begin()
            self.out = model.forward_backward(self.static, 0)
            segs[-1].capture_end()
            eng.on_group_done = None
            self.gopt = torch.cuda.CUDAGraph()
            self.gopt.capture_begin(pool=segs[0].pool())
            model.apply_optimizers()
            self.gopt.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        self.segs, self.split_at = segs, split_at

    def _eager_step(self, s):
        self.eng.on_group_done = self.hook
        out = self.model.forward_backward(self.static, s)
        self.eng.on_group_done = None
        self.finish()
        self.model.apply_optimizers()
        return out

    def step(self, s=0):
        if not self.use_graph:
            return self._eager_step(s)
        for g, k in zip(self.segs, self.split_at + [None]):
            g.replay()
            if k is not None:
                self.hook(k)
        self.finish()
        self.gopt.replay()
        return self.out


def run(args, rank, world, barrier_sync, time_kernel, cpu_baseline_fn=None):
    P, K, H, W = 16, 4, 256, 128
    arch = os.environ.get("CREID_BENCH_ARCH", "resnet50")      # resnet50_ibn_a: side measurement, not the headline config
    if os.environ.get("CREID_BENCH_CONFIG3", "0") == "1":      # side line: the training half of BASELINE configs[3]
        arch, P, H, W = "resnet50_ibn_a", 14, 320, 320         # (configs/320_resnet50_ibn_a.yml: 320 x 320, 14 x 4 images)
    f32 = os.environ.get("CREID_BENCH_DTYPE", "bf16") == "f32"   # side measurement: the exact-f32 parity mode
    model = make_model(arch=arch, dtype=torch.float32 if f32 else torch.bfloat16)
    if world > 1:
        # identical initial weights on every rank, then data-parallel gradient all-reduce over RCCL
        opt, _ = model.optimizers()
        dist.broadcast(opt.flat, 0)
        model.backbone.engine.weights_dirty = True      # the parameters are views of opt.flat
        dist.broadcast(model.center_loss.centers.data, 0)
        for b in model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, 0)
        model.grad_sync = parallel.make_grad_sync(world)
        overlap = os.environ.get("CREID_DDP_OVERLAP", "1") == "1"
    batches = [synthetic_batch(P, K, H, W, s, rank) for s in range(4)]
    use_graph = os.environ.get("CREID_NO_GRAPH", "0") != "1"
    split_graph = world > 1 or os.environ.get("CREID_SPLIT_GRAPH", "0") == "1"
    stepper = None
    if world > 1 and overlap:
        sx, sl = batches[0][0].clone(), batches[0][1].clone()
        try:
            stepper = DDPStepper(model, world, (sx, sl, batches[0][2], batches[0][3]), use_graph=use_graph)
        except Exception as e:  # noqa: BLE001  (capture problems must not cost the measurement: use the two-graph path)
            import sys
            print(f"[bench] overlapped data-parallel step unavailable ({type(e).__name__}: {e}); "
                  "falling back to graph A -> all-reduce -> graph B", file=sys.stderr)
            stepper = None
            model.backbone.engine.on_group_done = None
            model.grad_sync = parallel.make_grad_sync(world)
    if stepper is not None:
        def one_step(s):
            sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
            return stepper.step(s)
    elif use_graph:
        # The step is captured ONCE into hipGraphs; every timed step copies a fresh synthetic batch into the
        # static input buffers and replays.  Single GPU: one graph for the whole step.  Data parallel: graph A =
        # forward + losses + backward, then the RCCL all-reduce of the flat gradient buffer (eager, same stream),
        # then graph B = Adam + center SGD.
        sx, sl = batches[0][0].clone(), batches[0][1].clone()
        static = (sx, sl, batches[0][2], batches[0][3])
        sync = model.grad_sync
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for s in range(3):
                model.training_step(static, s)
        torch.cuda.current_stream().wait_stream(side)
        if not split_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gout = model.training_step(static, 0)

            def one_step(s):
                sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
                graph.replay()
                return gout
        else:
            model.grad_sync = None
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                gout = model.forward_backward(static, 0)
            if sync is not None:
                sync(model)                      # also fixes opt.grad_scale = 1/world before graph B is captured
            with torch.cuda.graph(gb, pool=ga.pool()):
                model.apply_optimizers()

            def one_step(s):
                sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
                ga.replay()
                if sync is not None:
                    sync(model)
                gb.replay()
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
        gflop_img = 76.0 if (H, W) == (320, 320) else R50_FWD_BWD_GFLOP_PER_IMG      # IBN-a 320x320 / R50 256x128
        tf_step = gflop_img * P * K / (ms * 1e-3) / 1e3
        res["step_mfma_frac"] = tf_step / MFMA_BF16_TFLOPS     # all conv FLOPs / whole-step time (incl. HBM-bound BN etc.)
        tf, ig_ms, slow, fast = igemm_roofline(P * K, H, W, time_kernel)
        res["roofline"] = {"kernel": "igemm_bf16_{dma,ws}_kernel (conv fwd + dgrad, 104 launches/step, real layer mix)",
                           "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": tf / MFMA_BF16_TFLOPS, "traffic": pmc_traffic("igemm_family"),
                           "mfma_busy_by_counter": pmc_field("igemm_family", "mfma_busy"),   # in-situ style: every launch, cold operands
                           "ms_per_step": ig_ms, "slowest_TFs": slow, "fastest_TFs": fast}
        res["roofline_hbm_stages"] = hbm_stage_rates(time_kernel, P * K, H, W)
        if cpu_baseline_fn is not None and world == 1:          # the CPU leg is reported at N=1 only
            res["cpu_baseline"] = cpu_baseline_fn(P, K, H, W)
        if world == 1 and arch == "resnet50" and not f32 and os.environ.get("CREID_BENCH_NO_EVAL", "0") != "1":
            del model
            torch.cuda.empty_cache()
            res["map_delta_bf16"] = map_delta_bf16()             # BASELINE metric (iii) on clustered synthetic identities
    return {"metric": "train_images_per_sec", "value": imgs / dt, "unit": "images/s",
            "ms_per_step": dt / args.steps * 1e3, "dtype": "f32" if f32 else "bf16",
            "config": {"workload": ("ResNet50" if arch == "resnet50" else arch) +
                                   f" {H}x{W} CTL training step: fwd+bwd, centroid-triplet + center + xent, "
                                   "Adam + center SGD (BASELINE configs[1])" if (H, W) == (256, 128) else
                                   f"{arch} {H}x{W} CTL training step, P={P} x K={K} (training half of BASELINE configs[3], side line)",
                       "P": P, "K": K, "global_batch": P * K * world, "num_classes": 751,
                       "parallelism": f"dp{world}" if world > 1 else "single"}, **res}
