"""Which 16-bit storage policy of the eval-mode forward would reach |delta mAP| <= 1e-4 against the fp32 forward?
("f16 MFMA inputs, residual trunk kept in fp32".)  Decided by EMULATION on the host, before any kernel
is written: the folded eval-mode forward (conv -> fp32 affine -> (+residual) -> ReLU -> round -> store) restated in
torch-CPU fp32 with an explicit rounding function at every place where the HIP forward rounds, one policy per run:

    f32        no rounding anywhere (the reference value)
    bf16, f16  what the HIP forward does today: every stored activation and every weight rounded to the 16-bit type
    f16_trunk  f16 as above, but a block's OUTPUT (the residual trunk) stays fp32 in memory; the next block's conv1 /
               downsample read it rounded to f16 (the MFMA operand), the residual add reads it unrounded
    f16_hilo   trunk32 + the trunk's MFMA operand as an f16 (hi, lo) pair: conv1 / downsample see the unrounded trunk
               (two MFMA passes over the same weights), c1 / c2 outputs stay f16
    f16_w32    f16 activations, weights NOT rounded (isolates the weight rounding's share)

Same clustered-identity recipe as bench_train.map_delta_bf16 (images from a CPU generator, so the absolute mAP differs
from the GPU recipe's; the deltas are what is read).  Uses oracle/ as a checker -- this is a measurement tool, not product.
    python tests/probes/f16_hi_emul.py [noise ...]      (default 0.3 0.6 0.9)"""
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import backbone_oracle as bo, reid_oracle as ro   # noqa: E402


def rounder(dt):
    return (lambda t: t) if dt is None else (lambda t: t.to(dt).float())


def fold(sd, name, eps=1e-5):
    s = sd[name + ".weight"] * torch.rsqrt(sd[name + ".running_var"] + eps)
    return s.view(1, -1, 1, 1), (sd[name + ".bias"] - sd[name + ".running_mean"] * s).view(1, -1, 1, 1)


def forward(x, sd, policy):
    act = rounder({"f32": None, "bf16": torch.bfloat16}.get(policy, torch.float16))
    wr = rounder(None) if policy in ("f32", "f16_w32") else act
    trunk32 = policy in ("f16_trunk", "f16_hilo")
    hilo = policy == "f16_hilo"

    def conv(a, name, stride=1, pad=0, bn=None, res=None, relu=True, keep32=False):
        y = F.conv2d(a, wr(sd[name + ".weight"]), stride=stride, padding=pad)
        s, t = fold(sd, bn)
        y = y * s + t
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return y if keep32 else act(y)

    # stem: the image operand is rounded too (image_pad writes the compute dtype); no stem ReLU in plain ResNet50
    y = conv(act(x), "conv1", 2, 3, "bn1", relu=False)
    y = F.max_pool2d(y, 3, 2, 1)
    for pre, _i, _p, s, ds, _ibn in bo.arch_spec("resnet50", 1):
        op = y if hilo else act(y)                  # what the block's first MFMAs read of the trunk
        o = conv(op, pre + ".conv1", bn=pre + ".bn1")
        o = conv(o, pre + ".conv2", s, 1, pre + ".bn2")
        # the downsample branch's normalised tensor is never stored (it rides in bn3's pass): no rounding of its own
        r = conv(op, pre + ".downsample.0", s, 0, pre + ".downsample.1", relu=False, keep32=True) if ds else y
        y = conv(o, pre + ".conv3", bn=pre + ".bn3", res=r, keep32=trunk32)
    return y.mean(dim=(2, 3))


def main():
    noises = [float(a) for a in sys.argv[1:]] or [0.3, 0.6, 0.9]
    torch.manual_seed(0)
    n_id, nq, ng, H, W = 96, 2, 6, 256, 128
    per = nq + ng
    sd = bo.make_state_dict("resnet50", 1, seed=1234)
    for k in sd:                                  # the product model's initialisation (random_init: BN gamma 1, beta 0, fresh statistics)
        if k.endswith(("bn1.weight", "bn2.weight", "bn3.weight", "downsample.1.weight", "running_var")):
            sd[k] = torch.ones_like(sd[k])
        elif k.endswith(("bn1.bias", "bn2.bias", "bn3.bias", "downsample.1.bias", "running_mean")):
            sd[k] = torch.zeros_like(sd[k])
    # kaiming-normal std sqrt(2 / fan_out) like the reference's random_init; BN statistics settled below
    neck = {"w": torch.ones(2048), "b": torch.zeros(2048), "rm": torch.zeros(2048), "rv": torch.ones(2048)}
    policies = ("f32", "bf16", "f16", "f16_w32", "f16_trunk", "f16_hilo")
    print("| noise | policy | mAP | delta mAP | rank-1 | rel. L2 error of the embeddings: mean / max | rank-1 flips | s |\n|---|---|---|---|---|---|---|---|")
    for noise in noises:
        gen = torch.Generator().manual_seed(0)
        base = torch.randn((n_id, 3, H // 16, W // 16), generator=gen)
        base = F.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
        x = base.repeat_interleave(per, 0) + noise * torch.randn((n_id * per, 3, H, W), generator=gen)
        pid = np.repeat(np.arange(n_id), per)
        slot = np.tile(np.arange(per), n_id)
        q_rows = np.nonzero(slot < nq)[0]; g_rows = np.nonzero(slot >= nq)[0]
        order = np.concatenate([q_rows, g_rows])
        pids = pid[order]
        cams = np.concatenate([np.zeros(len(q_rows), np.int64), np.ones(len(g_rows), np.int64)])
        s2 = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            for s in range(0, 512, 64):                              # settle the running statistics (training mode, fp32)
                _, f = bo.backbone_forward(x[s:s + 64], s2, training=True)
                F.batch_norm(f, neck["rm"], neck["rv"], neck["w"], neck["b"], True, 0.1, 1e-5)
        ref = None
        for pol in policies:
            t0 = time.time()
            out = []
            with torch.no_grad():
                for s in range(0, len(x), 64):
                    f = forward(x[s:s + 64], s2, pol)
                    out.append(F.batch_norm(f, neck["rm"], neck["rv"], neck["w"], neck["b"], False, 0.1, 1e-5))
            e = torch.cat(out)[torch.as_tensor(order)].contiguous()
            cmc, mAP, _, _ = ro.r1_map(e, pids, cams, len(q_rows))
            en = F.normalize(e, dim=1)
            top1 = (en[:len(q_rows)] @ en[len(q_rows):].T).argmax(1)
            if ref is None:
                ref = (en, mAP, top1)
            rel = (en - ref[0]).norm(dim=1)
            print(f"| {noise} | {pol} | {mAP:.6f} | {mAP - ref[1]:+.2e} | {float(cmc[0]):.4f} | {rel.mean().item():.2e} / {rel.max().item():.2e} | "
                  f"{int((top1 != ref[2]).sum())} | {time.time() - t0:.0f} |", flush=True)


if __name__ == "__main__":
    main()
