"""Generate golden input/output vectors by RUNNING THE REFERENCE on CPU in this container.

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference (read-only at /root/reference, pure Python) is imported under the
sys.modules stubs of tools/ref_import.py; only DATA (inputs + the reference's outputs)
is written to tests/golden/.  Inputs come from numpy seeded generators or closed-form
recipes and are stored alongside the outputs, so the fixtures are self-contained.
This script never runs on the GPU box (the reference does not travel).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
torch.manual_seed(0)


# ---------------------------------------------------------------------------
def make_cfg(ref, **over):
    """Reference-style cfg (AttrDict tree) with the reference defaults."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "creid_cfg", os.path.join(ROOT, "centroids-reid_amd", "config.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    c = m.get_cfg_defaults()

    def conv(d):
        a = ref_import.AttrDict()
        for k, v in d.items():
            a[k] = conv(v) if isinstance(v, dict) else v
        return a
    c = conv(c)
    c.MODEL.PRETRAINED = False
    c.SOLVER.OPTIMIZER_NAME = "Adam"
    for k, v in over.items():
        node = c
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return c


def gapped_features(nq, ng, D, seed, min_gap, iters=20000):
    """Unit-norm features whose per-query sorted distances have adjacent gaps >= min_gap
    (so that rank order is implementation-independent: SURVEY.md §7 'hard parts')."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((nq + ng, D)).astype(np.float32)
    for it in range(iters):
        fn = f / np.maximum(np.linalg.norm(f.astype(np.float64), axis=1, keepdims=True), 1e-12)
        q, g = fn[:nq], fn[nq:]
        d = ((q * q).sum(1)[:, None] + (g * g).sum(1)[None, :] - 2 * q @ g.T)
        order = np.argsort(d, axis=1)
        ds = np.take_along_axis(d, order, 1)
        gaps = np.diff(ds, axis=1)
        bad = np.argwhere(gaps < min_gap)
        if len(bad) == 0:
            return f, it
        qi, k = bad[0]
        j = order[qi, k + 1]
        f[nq + j] += (rng.standard_normal(D) * 0.05).astype(np.float32)
    raise RuntimeError("could not build a gapped fixture")


def gen_eval(ref, name, nq, ng, D, seed, n_pid, n_cam, min_gap, force_invalid=0, split_cams=False, slim=False):
    f, it = gapped_features(nq, ng, D, seed, min_gap)
    rng = np.random.default_rng(seed + 1)
    pids = rng.integers(0, n_pid, nq + ng).astype(np.int64)
    cams = rng.integers(0, n_cam, nq + ng).astype(np.int64)
    if split_cams:                           # no same-camera removals (kept length == ng)
        cams[:nq] = 0; cams[nq:] = 1
    for k in range(force_invalid):           # queries whose pid never appears in the gallery
        pids[k] = n_pid + 100 + k
    feats = torch.from_numpy(f)
    # --- the reference path: utils/reid_metric.py:112-135 then utils/eval_reid.py:25-92
    fn = torch.nn.functional.normalize(feats, dim=1, p=2)
    distmat = ref.reid_metric.get_euclidean(x=fn[:nq], y=fn[nq:])
    indices = np.argsort(distmat.numpy(), axis=1)
    cmc, mAP, topk, single = ref.eval_reid.eval_func(
        indices, pids[:nq], pids[nq:], cams[:nq], cams[nq:], 50, False)
    ds = np.take_along_axis(distmat.numpy().astype(np.float64), indices, 1)
    big = {} if slim else dict(feats_norm=fn.numpy(), distmat=distmat.numpy())
    np.savez_compressed(
        os.path.join(OUT, name), feats=f, pids=pids, camids=cams, num_query=np.int64(nq),
        indices=indices.astype(np.int64), **big,
        cmc=np.asarray(cmc, np.float32), mAP=np.float64(mAP), topk=np.asarray(topk, np.float64),
        single=np.asarray(single, np.float64), min_gap=np.float64(np.diff(ds, axis=1).min()))
    print(f"[{name}] nq={nq} ng={ng} D={D} repair_iters={it} min_gap={np.diff(ds, axis=1).min():.3e} "
          f"mAP={mAP:.6f} valid={len(single)}")


def gen_eval_centroids(ref, name, nq, ng, D, seed, n_pid):
    """modelling/bases.py:179-262 validation_create_centroids(respect_camids=False) + metric."""
    f, it = gapped_features(nq, ng, D, seed, 0.0)
    rng = np.random.default_rng(seed + 1)
    pids = np.concatenate([rng.integers(0, n_pid, nq), rng.integers(0, n_pid, ng)]).astype(np.int64)
    cams = np.concatenate([np.zeros(nq), np.ones(ng)]).astype(np.int64)
    cfg = make_cfg(ref)
    fake = types.SimpleNamespace(hparams=ref_import.AttrDict(num_query=nq))
    fake._calculate_centroids = ref.bases.ModelBase._calculate_centroids
    emb, labels, camids = ref.bases.ModelBase.validation_create_centroids(
        fake, torch.from_numpy(f), pids, cams, respect_camids=False)
    fn = torch.nn.functional.normalize(emb.float(), dim=1, p=2)
    distmat = ref.reid_metric.get_euclidean(x=fn[:nq], y=fn[nq:])
    indices = np.argsort(distmat.numpy(), axis=1)
    cmc, mAP, topk, single = ref.eval_reid.eval_func(
        indices, labels[:nq], labels[nq:], camids[:nq], camids[nq:], 50, False)
    np.savez_compressed(
        os.path.join(OUT, name), feats=f, pids=pids, camids=cams, num_query=np.int64(nq),
        cent_emb=emb.numpy(), cent_labels=np.asarray(labels, np.int64),
        cent_camids=np.asarray(camids, np.int64), cmc=np.asarray(cmc, np.float32),
        mAP=np.float64(mAP), topk=np.asarray(topk, np.float64))
    print(f"[{name}] centroids={emb.shape[0] - nq} mAP={mAP:.6f}")


# ---------------------------------------------------------------------------
class FeatStub(torch.nn.Module):
    """Stands in for Baseline: returns preset features so the reference's head arithmetic
    (train_ctl_model.py:59-179) runs on known inputs and exposes d(loss)/d(features)."""

    def __init__(self, feats):
        super().__init__()
        self.feats = torch.nn.Parameter(feats.clone())

    def forward(self, x):
        return None, self.feats * 1.0


def build_ref_model(ref, cfg, num_classes, D):
    cfg.MODEL.BACKBONE_EMB_SIZE = D
    model = ref.train_ctl_model.CTLModel(cfg, num_classes=num_classes, num_query=0)
    return model


def gen_heads(ref, name, P, K, D, C, seed, fakes=(), margin=0.5, steps=1):
    rng = np.random.default_rng(seed)
    # same-pid features share a base vector; scales chosen so that BOTH hinge branches
    # (active / inactive) occur in the query triplet and in the centroid rounds
    sc = float(np.sqrt(128.0 / D))
    base = (rng.standard_normal((P, D))).astype(np.float32)
    feats = ((rng.standard_normal((P * K, D)) * (0.3 * sc)).astype(np.float32)
             + np.repeat(base, K, 0) * np.float32(0.2 * sc))
    pid_vals = (np.arange(P) * 7) % C
    labels = np.repeat(pid_vals, K).astype(np.int64)
    is_real = np.ones(P * K, bool)
    for j in fakes:
        is_real[j] = False
        # the reference pads with a zero IMAGE; its feature is whatever the backbone gives
        feats[j] = (rng.standard_normal(D) * 0.25 * sc).astype(np.float32)
    centers0 = rng.standard_normal((C, D)).astype(np.float32)
    fc0 = (rng.standard_normal((C, D)) * 0.001).astype(np.float32)
    bn_w0 = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    bn_b0 = np.zeros(D, np.float32)

    cfg = make_cfg(ref)
    cfg.SOLVER.MARGIN = margin
    cfg.DATALOADER.NUM_INSTANCE = K
    model = build_ref_model(ref, cfg, C, D)
    model.backbone = FeatStub(torch.from_numpy(feats))
    with torch.no_grad():
        model.center_loss.centers.copy_(torch.from_numpy(centers0))
        model.fc_query.weight.copy_(torch.from_numpy(fc0))
        model.bn.weight.copy_(torch.from_numpy(bn_w0))
    opts = ref.solver.build_optimizer(model.named_parameters(), model.hparams)
    model.optimizers = lambda use_pl_optimizer=True: opts
    model.manual_backward = lambda loss, optimizer=None: loss.backward()
    model.trainer = types.SimpleNamespace(current_epoch=0)
    model.train()
    batch = (torch.zeros(P * K, 3, 8, 4), torch.from_numpy(labels),
             torch.zeros(P * K, dtype=torch.int64), torch.from_numpy(is_real))
    rec = {}
    for s in range(steps):
        out = model.training_step(batch, s)
        rec[f"s{s}_loss_total"] = np.float32(out["loss"].item())
        for n in model.losses_names:
            rec[f"s{s}_{n}"] = np.float32(model.losses_dict[n][-1])
        for k, v in out["other"].items():
            rec[f"s{s}_{k}"] = np.float32(v)
        if s == 0:
            rec["s0_grad_features"] = model.backbone.feats.grad.detach().numpy().copy()
            # NB: centers.grad was rescaled in place by 1/CENTER_LOSS_WEIGHT (train_ctl_model.py:157-158)
            rec["s0_grad_centers_scaled"] = model.center_loss.centers.grad.detach().numpy().copy()
            rec["s0_grad_fc"] = model.fc_query.weight.grad.detach().numpy().copy()
            rec["s0_grad_bn_w"] = model.bn.weight.grad.detach().numpy().copy()
        rec[f"s{s}_centers_after"] = model.center_loss.centers.detach().numpy().copy()
        rec[f"s{s}_fc_after"] = model.fc_query.weight.detach().numpy().copy()
        rec[f"s{s}_bn_w_after"] = model.bn.weight.detach().numpy().copy()
        rec[f"s{s}_bn_rm_after"] = model.bn.running_mean.numpy().copy()
        rec[f"s{s}_bn_rv_after"] = model.bn.running_var.numpy().copy()
        rec[f"s{s}_feats_after"] = model.backbone.feats.detach().numpy().copy()
    masks, _ = ref.bases.ModelBase.create_masks_train(torch.from_numpy(labels))
    rec["masks"] = masks.numpy()
    np.savez_compressed(os.path.join(OUT, name), feats=feats, labels=labels, is_real=is_real,
                        centers0=centers0, fc0=fc0, bn_w0=bn_w0, bn_b0=bn_b0, P=np.int64(P),
                        K=np.int64(K), C=np.int64(C), margin=np.float32(margin),
                        base_lr=np.float64(cfg.SOLVER.BASE_LR), **rec)
    print(f"[{name}] total={rec['s0_loss_total']:.6f} xent={rec['s0_query_xent']:.6f} "
          f"trip={rec['s0_query_triplet']:.6f} center={rec['s0_query_center']:.6f} "
          f"ctl={rec['s0_centroid_triplet']:.6f}")


def gen_losses(ref, name, N, D, C, seed):
    """Stand-alone TripletLoss (margin / soft-margin / mask), CenterLoss, CrossEntropyLabelSmooth."""
    rng = np.random.default_rng(seed)
    K = 4
    x = rng.standard_normal((N, D)).astype(np.float32)
    labels = np.repeat((np.arange(N // K) * 5) % C, K).astype(np.int64)
    mask = np.ones(N, bool); mask[[1, 6]] = False
    rec = dict(x=x, labels=labels, mask=mask)
    for tag, margin, m in (("m05", 0.5, None), ("soft", None, None), ("m05_mask", 0.5, mask)):
        xt = torch.from_numpy(x).requires_grad_(True)
        tl = ref.triplet_loss.TripletLoss(margin, "euclidean")
        loss, ap, an = tl(xt, torch.from_numpy(labels), mask=None if m is None else torch.from_numpy(m))
        loss.backward()
        rec[f"trip_{tag}_loss"] = np.float32(loss.item())
        rec[f"trip_{tag}_ap"] = ap.detach().numpy(); rec[f"trip_{tag}_an"] = an.detach().numpy()
        rec[f"trip_{tag}_grad"] = xt.grad.numpy().copy()
    dist = ref.triplet_loss.euclidean_dist(torch.from_numpy(x), torch.from_numpy(x))
    ap, an, pi, ni = ref.triplet_loss.hard_example_mining(dist, torch.from_numpy(labels), return_inds=True)
    rec["dist"] = dist.numpy(); rec["p_inds"] = pi.numpy(); rec["n_inds"] = ni.numpy()
    # center loss
    centers = rng.standard_normal((C, D)).astype(np.float32)
    cl = ref.center_loss.CenterLoss(C, D, use_gpu=False)
    with torch.no_grad():
        cl.centers.copy_(torch.from_numpy(centers))
    xt = torch.from_numpy(x).requires_grad_(True)
    l = cl(xt, torch.from_numpy(labels)); l.backward()
    rec.update(centers=centers, center_loss=np.float32(l.item()), center_grad_x=xt.grad.numpy().copy(),
               center_grad_c=cl.centers.grad.numpy().copy())
    # label-smoothed xent
    logits = (rng.standard_normal((N, C)) * 2).astype(np.float32)
    lt = torch.from_numpy(logits).requires_grad_(True)
    xe = ref.triplet_loss.CrossEntropyLabelSmooth(C, use_gpu=False)
    l = xe(lt, torch.from_numpy(labels)); l.backward()
    rec.update(logits=logits, xent_loss=np.float32(l.item()), xent_grad=lt.grad.numpy().copy())
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(f"[{name}] trip={rec['trip_m05_loss']:.6f} soft={rec['trip_soft_loss']:.6f} "
          f"center={rec['center_loss']:.4f} xent={rec['xent_loss']:.6f}")


# ---------------------------------------------------------------------------
def gen_backbone(ref, name, arch, B, H, W):
    from oracle import backbone_oracle as bo
    sd = bo.make_state_dict(arch, 1, seed=1234)
    if arch.endswith("_ibn_a"):
        net = getattr(ref.resnet_ibn_a, arch)(1)
    else:
        net = ref.resnet.ResNet(last_stride=1, block=ref.resnet.Bottleneck, layers=list(bo.ARCH_LAYERS[arch]))
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all(k.startswith("fc.") for k in missing.missing_keys), missing.missing_keys
    x = bo.synthetic_images(B, H, W, seed=7)
    coef = torch.from_numpy(np.random.default_rng(99).standard_normal((B, 2048)).astype(np.float32))
    rec = {}
    net.eval()
    with torch.no_grad():
        y = net(x)
        rec["eval_feat"] = y.mean(dim=(2, 3)).numpy()
        rec["eval_base_checksum"] = np.float64(y.double().sum().item())
    net.train()
    y = net(x)
    feat = y.mean(dim=(2, 3))
    rec["train_feat"] = feat.detach().numpy()
    (feat * coef).sum().backward()
    rec["bn1_rm"] = net.bn1.running_mean.numpy().copy(); rec["bn1_rv"] = net.bn1.running_var.numpy().copy()
    last = net.layer4[2].bn3
    rec["l4_bn3_rm"] = last.running_mean.numpy().copy(); rec["l4_bn3_rv"] = last.running_var.numpy().copy()
    rec["grad_conv1"] = net.conv1.weight.grad.numpy().copy()
    rec["grad_l4_conv3_slice"] = net.layer4[2].conv3.weight.grad[:16, :, 0, 0].numpy().copy()
    rec["grad_l1_conv2_slice"] = net.layer1[0].conv2.weight.grad[:8].numpy().copy()
    rec["grad_l2_ds_slice"] = net.layer2[0].downsample[0].weight.grad[:8, :, 0, 0].numpy().copy()
    rec["grad_bn1_w"] = net.bn1.weight.grad.numpy().copy(); rec["grad_bn1_b"] = net.bn1.bias.grad.numpy().copy()
    rec["grad_l3_bn2_w"] = net.layer3[1].bn2.weight.grad.numpy().copy()
    if arch.endswith("_ibn_a"):
        rec["grad_l1_in_w"] = net.layer1[0].bn1.IN.weight.grad.numpy().copy()
    gsum = 0.0
    for p in net.parameters():
        if p.grad is not None:
            gsum += p.grad.double().abs().sum().item()
    rec["grad_abs_sum"] = np.float64(gsum)
    np.savez_compressed(os.path.join(OUT, name), arch=np.array(arch), B=np.int64(B), H=np.int64(H),
                        W=np.int64(W), **rec)
    print(f"[{name}] eval_feat mean={rec['eval_feat'].mean():.5f} std={rec['eval_feat'].std():.5f} "
          f"train_feat std={rec['train_feat'].std():.5f} grad_abs_sum={gsum:.4e}")


def gen_full_step(ref, name="full_step_r50_p4k4_64x32", arch="resnet50", H=64, W=32, seed=77):
    """The reference's whole training_step (real ResNet50 + BNNeck + the four losses + both optimiser steps) on a small PK batch
    with one isReal = False sample, in fp32.  (Under torch.autocast(cpu) the step does not run: losses/triplet_loss.py:34's in-place
    addmm_ mixes the 16-bit features with fp32 terms, which only the CUDA autocast lists reconcile -- so the precision=16 anchor is
    the backbone recording of gen_backbone_autocast, where the 16-bit arithmetic lives.)  Weights / batch: those of
    tests/test_ctl_step_gpu.py::test_full_model_fp32_vs_oracle."""
    from oracle import backbone_oracle as bo
    P, K, C = 4, 4, 20
    sd = bo.make_state_dict(arch, 1, seed=seed)
    rng = np.random.default_rng(5)
    centers0 = torch.from_numpy(rng.standard_normal((C, 2048)).astype(np.float32)) * 0.3
    fc0 = torch.from_numpy((rng.standard_normal((C, 2048)) * 0.01).astype(np.float32))
    x = bo.synthetic_images(P * K, H, W, seed=3)
    labels = torch.from_numpy(np.repeat(np.arange(P) * 3 % C, K).astype(np.int64))
    is_real = torch.ones(P * K, dtype=torch.bool); is_real[6] = False
    rec = {}
    for tag, dt in (("f32", None),):
        cfg = make_cfg(ref)
        cfg.SOLVER.MARGIN = 0.5
        cfg.DATALOADER.NUM_INSTANCE = K
        cfg.MODEL.BACKBONE_EMB_SIZE = 2048
        cfg.MODEL.NAME = arch
        model = ref.train_ctl_model.CTLModel(cfg, num_classes=C, num_query=0)
        missing = model.backbone.base.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys, missing
        with torch.no_grad():
            model.center_loss.centers.copy_(centers0)
            model.fc_query.weight.copy_(fc0)
        opts = ref.solver.build_optimizer(model.named_parameters(), model.hparams)
        model.optimizers = lambda use_pl_optimizer=True, o=opts: o
        model.manual_backward = lambda loss, optimizer=None: loss.backward()
        model.trainer = types.SimpleNamespace(current_epoch=0)
        model.train()
        batch = (x, labels, torch.zeros(P * K, dtype=torch.int64), is_real)
        out = model.training_step(batch, 0)
        rec[f"{tag}_loss_total"] = np.float32(float(out["loss"]))
        for n in model.losses_names:
            rec[f"{tag}_{n}"] = np.float32(model.losses_dict[n][-1])
        print(f"[{name}] {tag}: total={rec[tag + '_loss_total']:.6f} " + " ".join(f"{n}={rec[tag + '_' + n]:.6f}" for n in model.losses_names))
        # three more steps on fresh batches of the same identities (trajectory: Adam + the centers' SGD + BatchNorm running
        # statistics feed back into the next step's losses)
        for st in range(1, 4):
            xs = bo.synthetic_images(P * K, H, W, seed=3 + st)
            out = model.training_step((xs, labels, torch.zeros(P * K, dtype=torch.int64), torch.ones(P * K, dtype=torch.bool)), st)
            rec[f"{tag}_s{st}_loss_total"] = np.float32(float(out["loss"]))
            for n in model.losses_names:
                rec[f"{tag}_s{st}_{n}"] = np.float32(model.losses_dict[n][-1])
            print(f"[{name}] step {st}: total={rec[f'{tag}_s{st}_loss_total']:.6f}")
        rec["l4_bn3_rm_after"] = model.backbone.base.layer4[2].bn3.running_mean.numpy().copy()
        rec["bn_rv_after"] = model.bn.running_var.numpy().copy()
        rec["centers_after"] = model.center_loss.centers.detach().numpy().copy()
        rec["conv1_after_slice"] = model.backbone.base.conv1.weight.detach().numpy()[:8].copy()
    np.savez_compressed(os.path.join(OUT, name), arch=np.array(arch), seed=np.int64(seed), P=np.int64(P), K=np.int64(K), C=np.int64(C),
                        H=np.int64(H), W=np.int64(W), **rec)


def gen_lr_schedule(ref, name="lr_schedule"):
    """The learning rate Adam actually steps with, epoch by epoch, under the reference's two mechanisms together: the warm-up
    written into training_step (train_ctl_model.py:41-49, overwrites every group's lr with lr_scale x BASE_LR) and the epoch
    scheduler of solver/build.py:50-63 (stepped once per epoch, as the Lightning trainer does), for both scheduler names and the
    default SOLVER values (BASE_LR, WARMUP_EPOCHS 10, LR_STEPS [40, 70], GAMMA 0.1, MAX_EPOCHS 120; config/defaults.py has no
    SOLVER.MIN_LR -- the cosine branch reads it from the user's yaml -- so it is set to 1e-7 here)."""
    rec = {}
    for sched in ("multistep_lr", "cosine_annealing"):
        for warm in (True, False):
            cfg = make_cfg(ref)
            cfg.SOLVER.LR_SCHEDULER_NAME = sched
            cfg.SOLVER.USE_WARMUP_LR = warm
            if sched == "cosine_annealing":
                cfg.SOLVER.MIN_LR = 1e-7
            w = torch.nn.Parameter(torch.zeros(4)); c = torch.nn.Parameter(torch.zeros(4))
            opts = ref.solver.build_optimizer([("w", w), ("center_loss.centers", c)], cfg)
            sch = ref.solver.build_scheduler(opts[0], cfg)
            used, center = [], []
            for epoch in range(int(cfg.SOLVER.MAX_EPOCHS)):
                if cfg.SOLVER.USE_WARMUP_LR and epoch < cfg.SOLVER.WARMUP_EPOCHS:       # train_ctl_model.py:41-49
                    lr_scale = min(1.0, float(epoch + 1) / float(cfg.SOLVER.WARMUP_EPOCHS))
                    for pg in opts[0].param_groups:
                        pg["lr"] = lr_scale * cfg.SOLVER.BASE_LR
                used.append(opts[0].param_groups[0]["lr"]); center.append(opts[1].param_groups[0]["lr"])
                w.grad = torch.ones(4); c.grad = torch.ones(4)
                opts[0].step(); opts[1].step()
                sch.step()
            key = f"{sched}_{'warm' if warm else 'nowarm'}"
            rec[key] = np.array(used, np.float64); rec[key + "_center"] = np.array(center, np.float64)
            print(f"[{name}] {key}: lr[0..12] = {np.round(np.array(used[:13]) * 1e4, 3).tolist()} x1e-4, lr[39..42] = {used[39:43]}, last {used[-1]:.3e}")
    np.savez_compressed(os.path.join(OUT, name), **rec)


def gen_backbone_autocast(ref, name, arch, B, H, W):
    """The reference's own modules under torch.autocast -- the op-level dtype policy of its `precision=16` trainer flag
    (utils/misc.py:111: Lightning native AMP = autocast around the step), executed here on the CPU (autocast(cpu) lowers the
    same convolutions / linears and leaves BatchNorm, ReLU, pooling and the residual adds to type promotion, like
    autocast(cuda)).  Same weights and images as gen_backbone(name without the suffix): the fp32 recording there is the anchor
    the 16-bit errors are measured against."""
    from oracle import backbone_oracle as bo
    sd = bo.make_state_dict(arch, 1, seed=1234)
    if arch.endswith("_ibn_a"):
        net = getattr(ref.resnet_ibn_a, arch)(1)
    else:
        net = ref.resnet.ResNet(last_stride=1, block=ref.resnet.Bottleneck, layers=list(bo.ARCH_LAYERS[arch]))
    net.load_state_dict(sd, strict=False)
    x = bo.synthetic_images(B, H, W, seed=7)
    rec = {}
    for tag, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        net.load_state_dict(sd, strict=False)                      # running statistics back to the initial ones
        net.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=dt):
            y = net(x)
        rec[f"eval_dtype_{tag}"] = np.array(str(y.dtype))
        rec[f"eval_feat_{tag}"] = y.float().mean(dim=(2, 3)).numpy()
        net.train()
        with torch.no_grad(), torch.autocast("cpu", dtype=dt):
            y = net(x)
        rec[f"train_feat_{tag}"] = y.float().mean(dim=(2, 3)).numpy()
        print(f"[{name}] {tag}: out dtype {y.dtype}")
    np.savez_compressed(os.path.join(OUT, name), arch=np.array(arch), B=np.int64(B), H=np.int64(H), W=np.int64(W), **rec)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_import.load()
    which = [a for a in sys.argv[1:] if a not in ("sampler", "ckpt", "camsets", "inference", "surface_r2", "ibn320", "streamed", "transforms", "deep", "config", "basic", "autocast")] or (["eval", "losses", "heads", "backbone"] if len(sys.argv) == 1 else [])
    if "eval" in which:
        gen_eval(ref, "eval_small", 32, 256, 64, 11, n_pid=24, n_cam=4, min_gap=2e-5, force_invalid=2)
        gen_eval(ref, "eval_d2048", 24, 200, 2048, 12, n_pid=25, n_cam=6, min_gap=1e-5, force_invalid=1, slim=True)
        gen_eval(ref, "eval_tiny_gallery", 8, 40, 32, 13, n_pid=5, n_cam=3, min_gap=1e-4, split_cams=True)
        gen_eval_centroids(ref, "eval_centroids", 40, 300, 128, 14, n_pid=30)
    if "losses" in which:
        gen_losses(ref, "losses_n64_d128", 64, 128, 751, 21)
        gen_losses(ref, "losses_n32_d2048", 32, 2048, 40, 22)
    if "heads" in which:
        gen_heads(ref, "heads_p16k4_d128", 16, 4, 128, 200, 31, steps=2)
        gen_heads(ref, "heads_p16k4_d128_fake1", 16, 4, 128, 60, 32, fakes=(5,))
        gen_heads(ref, "heads_p16k4_d128_fake2", 16, 4, 128, 60, 33, fakes=(8, 9, 30))
        gen_heads(ref, "heads_p8k4_d2048", 8, 4, 2048, 24, 34)
    if "autocast" in sys.argv[1:]:
        gen_backbone_autocast(ref, "backbone_r50_autocast_2x256x128", "resnet50", 2, 256, 128)
        gen_backbone_autocast(ref, "backbone_r50ibn_autocast_2x64x64", "resnet50_ibn_a", 2, 64, 64)
        gen_full_step(ref)
        gen_lr_schedule(ref)
        gen_full_step(ref, "full_step_r50ibn_p4k4_64x64", "resnet50_ibn_a", 64, 64, seed=79)
    if "backbone" in which:
        gen_backbone(ref, "backbone_r50_2x256x128", "resnet50", 2, 256, 128)
        gen_backbone(ref, "backbone_r50ibn_2x64x64", "resnet50_ibn_a", 2, 64, 64)


if __name__ == "__main__":
    main()


def gen_sampler():
    """PK sampler + per-PID dataset of the reference on a synthetic identity table."""
    import importlib, random, json
    ref_import.install_stubs(); ref_import.ref_path_first()
    smod = importlib.import_module("datasets.samplers.distributed_pids_sampler")
    rng = np.random.default_rng(41)
    counts = rng.integers(2, 11, 60)
    table = {int(p): [(f"img{p}_{i}", int(p), int(rng.integers(0, 6)), int(1000 * p + i)) for i in range(int(c))]
             for p, c in enumerate(counts)}
    rec = {"counts": counts.astype(np.int64)}
    for world in (1, 2):
        for rank in range(world):
            s = smod.RandomIdentitySampler({p: [t[3] for t in v] for p, v in table.items()}, 4, 4, world, rank)
            for ep in (0, 1):
                s.set_epoch(ep)
                rec[f"w{world}_r{rank}_e{ep}"] = np.asarray([int(x) for x in s], np.int64)
                rec[f"w{world}_r{rank}_e{ep}_len"] = np.int64(len(s))
    bases = importlib.import_module("datasets.bases")
    for resample in (False, True):
        ds = bases.BaseDatasetLabelledPerPid({p: list(v) for p, v in table.items()}, None, 4, resample)
        ds.prepare_img = lambda path: torch.full((1, 2, 2), float(int(path[3:].split("_")[0]) * 100 + int(path.split("_")[1])))
        random.seed(5); np.random.seed(5)
        rows = []
        for pid in (0, 3, 7, 3, 11, 20):
            if len(ds.samples[pid]) <= 1:
                continue
            out = ds[pid]
            rows.append([[float(t[0].flatten()[0]), t[1], t[2], t[3], int(t[4])] for t in out])
        rec[f"items_resample{int(resample)}"] = np.asarray(rows, np.float64)
    np.savez_compressed(os.path.join(OUT, "sampler"), table=np.array(json.dumps({str(k): v for k, v in table.items()})), **rec)
    print("[sampler] sequences", {k: len(v) for k, v in rec.items() if k.startswith("w") and not k.endswith("len")})


if __name__ == "__main__" and "sampler" in sys.argv[1:]:
    gen_sampler()


def gen_ckpt_keys():
    """state_dict key/shape layout of the reference's CTLModel (R50 and R50-IBN-a) -> checkpoint contract."""
    ref = ref_import.load()
    rec = {}
    for arch in ("resnet50", "resnet50_ibn_a", "resnet101", "resnet152", "resnet101_ibn_a"):    # (the last three: round 5)
        cfg = make_cfg(ref)
        cfg.MODEL.NAME = arch
        m = ref.train_ctl_model.CTLModel(cfg, num_classes=751, num_query=10)
        sd = m.state_dict()
        rec[f"{arch}_keys"] = np.array(list(sd.keys()))
        rec[f"{arch}_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "ckpt_keys"), **rec)
    print("[ckpt_keys]", {k: len(v) for k, v in rec.items()})


if __name__ == "__main__" and "ckpt" in sys.argv[1:]:
    gen_ckpt_keys()


def gen_camsets():
    """Camera-aware centroid evaluation (MODEL.KEEP_CAMID_CENTROIDS): modelling/bases.py:179-262 with
    respect_camids=True, then utils/eval_reid.py:25-92 with respect_camids=True."""
    ref = ref_import.load()
    nq, ng, D, n_pid, n_cam = 200, 3000, 32, 120, 5
    f, _ = gapped_features(nq, ng, D, 51, 0.0)
    rng = np.random.default_rng(52)
    pids = rng.integers(0, n_pid, nq + ng).astype(np.int64)
    cams = rng.integers(0, n_cam, nq + ng).astype(np.int64)
    fake = types.SimpleNamespace(hparams=ref_import.AttrDict(num_query=nq))
    fake._calculate_centroids = ref.bases.ModelBase._calculate_centroids
    emb, labels, camids = ref.bases.ModelBase.validation_create_centroids(fake, torch.from_numpy(f), pids, cams,
                                                                          respect_camids=True)
    fn = torch.nn.functional.normalize(emb.float(), dim=1, p=2)
    distmat = ref.reid_metric.get_euclidean(x=fn[:nq], y=fn[nq:])
    indices = np.argsort(distmat.numpy(), axis=1, kind="stable")
    g_sets = np.empty(len(camids) - nq, dtype=object)
    for i, cs in enumerate(camids[nq:]):
        g_sets[i] = list(cs)
    q_c = np.asarray([c[0] for c in camids[:nq]])
    cmc, mAP, topk, single = ref.eval_reid.eval_func(indices, labels[:nq], labels[nq:], q_c, g_sets, 50, True)
    ncent = emb.shape[0] - nq
    sets_flat = np.full((ncent, n_cam), -1, np.int64)
    for i, cs in enumerate(camids[nq:]):
        sets_flat[i, :len(cs)] = cs
    np.savez_compressed(os.path.join(OUT, "eval_camsets"), feats=f, pids=pids, camids=cams, num_query=np.int64(nq),
                        cent_emb=emb.numpy(), cent_labels=np.asarray(labels, np.int64), cent_camsets=sets_flat,
                        indices=indices.astype(np.int64), cmc=np.asarray(cmc, np.float32), mAP=np.float64(mAP),
                        topk=np.asarray(topk, np.float64), single=np.asarray(single, np.float64))
    print(f"[eval_camsets] centroids={ncent} mAP={mAP:.6f} valid={len(single)}")


if __name__ == "__main__" and "camsets" in sys.argv[1:]:
    gen_camsets()


def gen_inference():
    """Inference / similarity search: the reference's create_pid_path_index + calculate_centroids
    (inference/inference_utils.py:134-159) on synthetic embeddings, and the body of inference/get_similar.py:97-125
    (F.normalize -> get_dist_func('euclidean') -> np.argsort -> top-k) driven with the reference's own functions."""
    import importlib
    ref_import.install_stubs(); ref_import.ref_path_first()
    iu = importlib.import_module("inference.inference_utils")
    rm = importlib.import_module("utils.reid_metric")
    nq, ng, D, topk = 24, 400, 64, 15
    f, _ = gapped_features(nq, ng, D, 61, 2e-5)
    rng = np.random.default_rng(62)
    gal_pid = rng.integers(0, 37, ng)
    gpaths = np.array([f"gallery/{p:04d}_c{rng.integers(1, 7)}_{i:05d}.jpg" for i, p in enumerate(gal_pid)])
    qpaths = np.array([f"query/{i:04d}.jpg" for i in range(nq)])
    index = iu.create_pid_path_index(list(gpaths), lambda p: p.split("/")[-1].split("_")[0])
    cents, keys = iu.calculate_centroids(f[nq:], index)
    q = torch.nn.functional.normalize(torch.from_numpy(f[:nq]), dim=1, p=2)
    g = torch.nn.functional.normalize(torch.from_numpy(f[nq:]), dim=1, p=2)
    distmat = rm.get_dist_func("euclidean")(x=q, y=g).cpu().numpy()
    indices = np.argsort(distmat, axis=1)[:, :topk]
    dist_sel = np.take_along_axis(distmat, indices, 1)
    np.savez_compressed(os.path.join(OUT, "inference"), feats=f, num_query=np.int64(nq), topk=np.int64(topk),
                        gallery_paths=gpaths, query_paths=qpaths, index_keys=np.array(list(index.keys())),
                        index_sizes=np.array([len(v) for v in index.values()], np.int64),
                        index_flat=np.concatenate([np.asarray(v, np.int64) for v in index.values()]),
                        centroids=np.asarray(cents, np.float32), centroid_keys=keys,
                        indices=indices.astype(np.int64), distances=dist_sel.astype(np.float32))
    print(f"[inference] pids={len(index)} centroids={cents.shape} topk indices {indices.shape}")


if __name__ == "__main__" and "inference" in sys.argv[1:]:
    gen_inference()


def gen_surface_r2():
    """Round-2 boundary surface: create_masks_train on ragged / non-contiguous label vectors, TripletLoss with
    dist_func='cosine' and normalize_feature=True, euclidean_dist / cosine_dist between two DIFFERENT row sets,
    hard_example_mining on a given matrix -- all recorded from the reference's own functions."""
    ref = ref_import.load()
    tl = ref.triplet_loss
    rng = np.random.default_rng(2024)
    rec = {}
    label_sets = [np.array([3, 3, 3, 3, 9, 9, 9, 9, 1, 1, 1, 1]),            # regular P x K
                  np.array([5, 5, 5, 7, 7, 2, 2, 2, 2, 8]),                  # ragged, contiguous
                  np.array([4, 1, 4, 2, 1, 4, 2, 2, 2, 9]),                  # ragged, interleaved
                  np.array([6, 6, 0, 0, 0, 0, 3]),                           # first PID short (cumsum[-1] quirk)
                  np.array([11])]
    for i, lab in enumerate(label_sets):
        m, lists = ref.bases.ModelBase.create_masks_train(torch.from_numpy(lab))
        rec[f"mask{i}_labels"] = lab.astype(np.int64)
        rec[f"mask{i}_masks"] = m.numpy()
        rec[f"mask{i}_lists_flat"] = np.concatenate([np.asarray(v, np.int64) for v in lists])
        rec[f"mask{i}_lists_len"] = np.asarray([len(v) for v in lists], np.int64)
    rec["n_mask_sets"] = np.int64(len(label_sets))
    N, D, K = 32, 256, 4
    x = rng.standard_normal((N, D)).astype(np.float32) * 1.7
    labels = np.repeat(np.arange(N // K) * 3, K).astype(np.int64)
    mask = np.ones(N, bool); mask[[2, 9, 30]] = False
    rec.update(x=x, labels=labels, mask=mask)
    for tag, margin, dist, norm, m in (("cos_m05", 0.5, "cosine", False, None), ("cos_soft", None, "cosine", False, None),
                                       ("cos_m05_mask", 0.5, "cosine", False, mask), ("euc_norm", 0.5, "euclidean", True, None),
                                       ("cos_norm", 0.3, "cosine", True, None)):
        xt = torch.from_numpy(x).requires_grad_(True)
        loss, ap, an = tl.TripletLoss(margin, dist)(xt, torch.from_numpy(labels), normalize_feature=norm,
                                                    mask=None if m is None else torch.from_numpy(m))
        loss.backward()
        rec[f"trip_{tag}_loss"] = np.float32(loss.item())
        rec[f"trip_{tag}_ap"] = ap.detach().numpy(); rec[f"trip_{tag}_an"] = an.detach().numpy()
        rec[f"trip_{tag}_grad"] = xt.grad.numpy().copy()
    y = rng.standard_normal((20, D)).astype(np.float32)
    w = rng.standard_normal((N, 20)).astype(np.float32)
    for name, fn in (("euc", tl.euclidean_dist), ("cos", tl.cosine_dist)):
        xt = torch.from_numpy(x).requires_grad_(True); yt = torch.from_numpy(y).requires_grad_(True)
        d = fn(xt, yt)
        (d * torch.from_numpy(w)).sum().backward()
        rec[f"xy_{name}_dist"] = d.detach().numpy(); rec[f"xy_{name}_gx"] = xt.grad.numpy().copy()
        rec[f"xy_{name}_gy"] = yt.grad.numpy().copy()
    rec.update(y=y, w=w)
    dm = np.abs(rng.standard_normal((N, N))).astype(np.float32)
    ap, an, pi, ni = tl.hard_example_mining(torch.from_numpy(dm), torch.from_numpy(labels), return_inds=True)
    rec.update(mine_dist=dm, mine_ap=ap.numpy(), mine_an=an.numpy(), mine_pi=pi.numpy(), mine_ni=ni.numpy())
    np.savez_compressed(os.path.join(OUT, "surface_r2"), **rec)
    print(f"[surface_r2] cos_m05={rec['trip_cos_m05_loss']:.6f} cos_soft={rec['trip_cos_soft_loss']:.6f} "
          f"euc_norm={rec['trip_euc_norm_loss']:.6f} cos_norm={rec['trip_cos_norm_loss']:.6f}")


if __name__ == "__main__" and "surface_r2" in sys.argv[1:]:
    gen_surface_r2()


if __name__ == "__main__" and "deep" in sys.argv[1:]:
    # the deeper Bottleneck variants of MODEL.NAME (modelling/baseline.py:73-81): one small recording each
    for _arch, _tag in (("resnet101", "r101"), ("resnet152", "r152"), ("resnet101_ibn_a", "r101ibn")):
        gen_backbone(ref_import.load(), f"backbone_{_tag}_2x64x64", _arch, 2, 64, 64)


if __name__ == "__main__" and "ibn320" in sys.argv[1:]:
    # BASELINE configs[3] input size: ResNet50-IBN-a at 320 x 320 (20 x 20 final maps), batch 2
    gen_backbone(ref_import.load(), "backbone_r50ibn_2x320x320", "resnet50_ibn_a", 2, 320, 320)


def gen_transforms():
    """Input transforms: the reference's own RandomErasing (datasets/transforms/random_erasing.py, imported from the read-only
    reference tree; it needs only `math` and `random`) applied to normalised tensors made with torch's CPU ops from random uint8
    images under explicit flip / crop draws.  torchvision is absent from this image, so flip / pad / crop / ToTensor / Normalize
    are spelled with the torch / numpy ops torchvision documents (see oracle/transforms_oracle.py: PARITY UNPINNED for those);
    the erasing draws and the erased tensor are the reference's."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location("ref_random_erasing", os.path.join(ref_import.REF, "datasets", "transforms", "random_erasing.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean, std, pad = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], 10
    rng = np.random.default_rng(71)
    rec = {"mean": np.array(mean, np.float64), "std": np.array(std, np.float64), "pad": np.int64(pad)}
    cases = [(64, 32, 1.0, s) for s in range(5)] + [(64, 32, 0.5, s) for s in (10, 12, 15)] + [(256, 128, 1.0, 3), (33, 17, 1.0, 5)]
    for i, (H, W, prob, seed) in enumerate(cases):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        flip, top, left = int(rng.integers(0, 2)), int(rng.integers(0, 2 * pad + 1)), int(rng.integers(0, 2 * pad + 1))
        im = img[:, ::-1] if flip else img
        im = np.pad(im, ((pad, pad), (pad, pad), (0, 0)))[top:top + H, left:left + W]
        t = torch.from_numpy(np.ascontiguousarray(im)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)      # ToTensor
        t = t.clone().sub_(torch.tensor(mean)[:, None, None]).div_(torch.tensor(std)[:, None, None])                 # Normalize
        before = t.clone()
        random.seed(seed)
        out = mod.RandomErasing(probability=prob, mean=mean)(t)
        state_after = random.random()                 # the next value of the stream: pins the NUMBER of draws consumed
        rec[f"c{i}_img"] = img; rec[f"c{i}_draw"] = np.array([flip, top, left, seed], np.int64); rec[f"c{i}_prob"] = np.float64(prob)
        rec[f"c{i}_out"] = out.numpy().copy(); rec[f"c{i}_next"] = np.float64(state_after)
        rec[f"c{i}_changed"] = np.int64(int((out != before).sum()))
    rec["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "transforms"), **rec)
    print(f"[transforms] {len(cases)} cases, erased elements: {[int(rec[f'c{i}_changed']) for i in range(len(cases))]}")


if __name__ == "__main__" and "transforms" in sys.argv[1:]:
    gen_transforms()


def gen_config_keys():
    """The reference's whole configuration tree (config/defaults.py:13-181: every key and its default), flattened to dotted
    names, as DATA: tests/golden/config_keys.json.  tests/test_checkpoint_cpu.py compares the product's config.get_cfg_defaults()
    against it key by key (the drop-in contract of north_star: "config/defaults.py keys ... stay intact")."""
    import importlib
    import json
    ref_import.install_stubs(); ref_import.ref_path_first()
    for m in [k for k in sys.modules if k == "config" or k.startswith("config.")]:
        del sys.modules[m]
    defaults = importlib.import_module("config.defaults")
    root = defaults._C

    def flat(node, prefix, out):
        for k, v in node.items():
            if isinstance(v, dict):
                flat(v, prefix + k + ".", out)
            else:
                out[prefix + k] = list(v) if isinstance(v, tuple) else v
        return out
    tree = flat(root, "", {})
    with open(os.path.join(OUT, "config_keys.json"), "w") as f:
        json.dump({"source": "config/defaults.py _C of the reference, flattened (tuples as lists)", "keys": tree}, f, indent=1, sort_keys=True)
    print(f"config_keys.json: {len(tree)} keys")


if __name__ == "__main__" and "config" in sys.argv[1:]:
    gen_config_keys()


def gen_backbone_basic(ref, name, arch, B, H, W):
    """resnet18 / resnet34 (modelling/baseline.py:56-65: ResNet(block=BasicBlock, layers=...), 512-wide embedding): the
    reference's own module on the oracle's PCG64-seeded weights -- eval / train embeddings, running statistics, gradients."""
    from oracle import backbone_oracle as bo
    sd = bo.make_state_dict(arch, 1, seed=1234)
    net = ref.resnet.ResNet(last_stride=1, block=ref.resnet.BasicBlock, layers=list(bo.ARCH_LAYERS[arch]))
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    x = bo.synthetic_images(B, H, W, seed=7)
    coef = torch.from_numpy(np.random.default_rng(99).standard_normal((B, 512)).astype(np.float32))
    rec = {}
    net.eval()
    with torch.no_grad():
        y = net(x)
        rec["eval_feat"] = y.mean(dim=(2, 3)).numpy()
    net.train()
    y = net(x)
    feat = y.mean(dim=(2, 3))
    rec["train_feat"] = feat.detach().numpy()
    (feat * coef).sum().backward()
    last = net.layer4[-1].bn2
    rec["l4_bn2_rm"] = last.running_mean.numpy().copy(); rec["l4_bn2_rv"] = last.running_var.numpy().copy()
    rec["grad_conv1"] = net.conv1.weight.grad.numpy().copy()
    rec["grad_l4_conv2_slice"] = net.layer4[-1].conv2.weight.grad[:8].numpy().copy()
    rec["grad_l1_conv1_slice"] = net.layer1[0].conv1.weight.grad[:8].numpy().copy()
    rec["grad_l2_ds"] = net.layer2[0].downsample[0].weight.grad[:, :, 0, 0].numpy().copy()
    rec["grad_l3_bn1_w"] = net.layer3[1].bn1.weight.grad.numpy().copy()
    gsum = sum(p.grad.double().abs().sum().item() for p in net.parameters() if p.grad is not None)
    rec["grad_abs_sum"] = np.float64(gsum)
    rec["n_keys"] = np.int64(len(net.state_dict()))
    np.savez_compressed(os.path.join(OUT, name), arch=np.array(arch), B=np.int64(B), H=np.int64(H), W=np.int64(W), **rec)
    print(f"[{name}] eval_feat std={rec['eval_feat'].std():.5f} train_feat std={rec['train_feat'].std():.5f} grad_abs_sum={gsum:.4e}")


if __name__ == "__main__" and "basic" in sys.argv[1:]:
    _ref = ref_import.load()
    gen_backbone_basic(_ref, "backbone_r18_2x128x64", "resnet18", 2, 128, 64)
    gen_backbone_basic(_ref, "backbone_r34_2x128x64", "resnet34", 2, 128, 64)
