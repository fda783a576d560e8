"""Measure per-shape launch plans on the GPU box and write centroids-reid_amd/tuned_plans.json.

    python tools/tune_plans.py [--batch 64 --h 256 --w 128]

For every distinct convolution of the ResNet50 layer mix at the benchmark size: the weight gradient is timed over
(tile, split) candidates (partial tiles + 0.4 x the stand-alone reduce: the reduction rides in the next launch in the
training schedule and costs ~40 % of its stand-alone time there), the forward / data gradient over (N tile, ring
depth) candidates.  A plan is recorded only when it beats the built-in rule by > 3 % (min of repeated measurements)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CREID_TUNED_PLANS"] = "0"                       # measure against the built-in rules
from bench import time_kernel                                # noqa: E402
from centroids_reid_amd import _lib as L, layers as ly       # noqa: E402
from centroids_reid_amd.bench_train import conv_shapes       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--h", type=int, default=256); ap.add_argument("--w", type=int, default=128)
ap.add_argument("--out", default=os.path.join(ROOT, "centroids-reid_amd", "tuned_plans.json"))
ap.add_argument("--merge", default=None, help="existing plan file whose entries (other shapes) are kept")
ap.add_argument("--fwd-only", action="store_true",
                help="eval-mode shapes (embedding batch sizes): forward with the folded BatchNorm epilogue only, incl. the "
                     "256-row tile variant (plan kind 3)")
ap.add_argument("--wgrad-only", action="store_true", help="re-measure the weight-gradient plans only (forward / data-gradient entries of --merge are kept)")
ap.add_argument("--pp-only", action="store_true",
                help="measure only the all-waves-multiply persistent kernel (conv_pipe.hip, plan kind 5) against the plans of --merge "
                     "(which stay registered as the baseline): training forward, or with --fwd-only the folded eval-mode forward")
args = ap.parse_args()
lib = L.lib()
B = args.batch
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 40, 48, 64, 96, 128, 192, 256, 384]


def t_us(fn, reps=2, iters=10):
    return min(time_kernel(fn, iters) for _ in range(reps)) * 1e3


PP_VARIANTS = [(bm // 128) | ((bn // 128) << 2) | (kph << 4) | (mode << 8)
               for (bm, bn) in ((256, 256), (128, 256), (256, 128), (128, 128)) for (kph, mode) in ((1, 0), (2, 0), (1, 2))]


def register(entries):
    for e in entries:
        lib.creid_tune_set(e["kind"], *e["key"], *e["plan"])


merge_plans = json.load(open(args.merge)).get("plans", []) if (args.merge and os.path.exists(args.merge)) else []
plans, log = [], []
tuned_keys = set()                                          # every (kind, key) measured in this run
seen = {}
for cin, cout, k, s, h, w in conv_shapes(B, args.h, args.w):
    key = (cin, cout, k, s, h, w)
    seen[key] = seen.get(key, 0) + 1
for (cin, cout, k, s, h, w), cnt in seen.items():
    pad = k // 2
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    y = ly.conv2d_fwd(x, krsc, s, pad)
    oh, ow = y.shape[1], y.shape[2]
    d, _, _ = ly.conv_desc(B, h, w, cin, cout, k, s, pad)
    M, K = B * oh * ow, k * k * cin
    dw = torch.zeros((cout, cin, k, k), device="cuda")
    name = f"{cin}->{cout} k{k} s{s} {h}x{w}"

    if args.pp_only:
        ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
        res_t = torch.randn_like(y) if (k == 1 and cout == 4 * cin) else None
        # COLD operands, as inside a captured forward: the launches of one timing cycle through R copies of the input (and of the
        # weights and the residual), R x input >= 64 MB = twice the L2s -- ten launches on ONE warm input flattered the kernels
        # with a short prefetch distance (profiles/r04_plan_validation.md)
        R = max(1, min(8, -(-64 * 2 ** 20 // (x.numel() * 2))))
        xs = [x] + [x.clone() for _ in range(R - 1)]
        ks = [krsc] + [krsc.clone() for _ in range(R - 1)]
        rs = [res_t] + [res_t.clone() if res_t is not None else None for _ in range(R - 1)]
        ctr = [0]
        if args.fwd_only:
            def fn():
                i = ctr[0] % R; ctr[0] += 1
                return ly.conv2d_fwd_affine(xs[i], ks[i], s, pad, ss, rs[i], True)
            key = (M, cout, K, (s << 1) | 8)
        else:
            def fn():
                i = ctr[0] % R; ctr[0] += 1
                return ly.conv2d_fwd(xs[i], ks[i], s, pad, with_stats=True)
            key = (M, cout, K, s << 1)
        if cout % 128 or K < 256:
            continue
        lib.creid_tune_clear()
        register([e for e in merge_plans if not (e["kind"] == 1 and e["plan"][2] == 5)])
        base = t_us(fn)
        best = (base, None)
        for v in PP_VARIANTS:
            if cout % (((v >> 2) & 3) * 128):
                continue
            lib.creid_tune_set(1, *key, v, 0, 5)
            sc = t_us(fn)
            if sc < best[0]:
                best = (sc, (v, 0, 5))
        lib.creid_tune_clear()
        tuned_keys.add((1, tuple(key)))
        if best[1] is not None and best[0] < 0.97 * base:
            plans.append({"kind": 1, "key": list(key), "plan": list(best[1]), "us": round(best[0], 2), "rule_us": round(base, 2),
                          "layer": f"{'fwd-eval' if args.fwd_only else 'fwd'} {name} B={B} (kind 5 = conv_pipe.hip, variant {hex(best[1][0])})"})
        log.append(f"pp {name:28s} x{cnt} plan/rule {base:6.1f}  best {best[0]:6.1f} {best[1] and hex(best[1][0])}")
        print(log[-1], flush=True)
        continue

    if args.fwd_only:
        ss = torch.stack([torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1]).contiguous()
        res_t = torch.randn_like(y) if (k == 1 and cout == 4 * cin) else None          # conv3: the block's residual rides in
        fn = lambda: ly.conv2d_fwd_affine(x, krsc, s, pad, ss, res_t, True)
        key = (M, cout, K, s << 1)
        lib.creid_tune_clear()
        tuned_keys.add((1, tuple(key)))
        base = t_us(fn)
        best = (base, None)
        for bn in (64, 128):
            if cout % bn:
                continue
            for st in (2, 3, 4):
                if bn == 128 and st == 4:
                    continue
                for kind in (0, 1, 3, 4):
                    if kind == 3 and (bn != 128 or st == 4 or (M + 255) // 256 * (cout // 128) < 256):
                        continue
                    if kind == 1 and st > 3:
                        continue
                    if kind == 4 and (bn != 128 or not (k == 1 and s == 1 and cin in (64, 128, 256))):   # second persistent 1x1 kernel; st = slab cap
                        continue
                    lib.creid_tune_set(1, *key, bn, st, kind)
                    sc = t_us(fn)
                    if sc < best[0]:
                        best = (sc, (bn, st, kind))
        lib.creid_tune_clear()
        if best[1] is not None and best[0] < 0.97 * base:
            plans.append({"kind": 1, "key": list(key), "plan": list(best[1]), "us": round(best[0], 2), "rule_us": round(base, 2),
                          "layer": f"fwd-eval {name} B={B}"})
        log.append(f"fwd-eval {name:28s} x{cnt} rule {base:6.1f}  best {best[0]:6.1f} {best[1]}")
        print(log[-1], flush=True)
        continue

    # ---------------- weight gradient
    def wgrad_score():
        nbytes = lib.creid_conv2d_wgrad_workspace_bytes(C.byref(d), L.BF16)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        tp = t_us(lambda: L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(x), L.ptr(y), L.ptr(ws), nbytes, L.BF16, L.stream()), "p"))
        tr = t_us(lambda: L.check(lib.creid_conv2d_wgrad_reduce(C.byref(d), L.ptr(dw), 0, L.ptr(ws), nbytes, L.BF16, L.stream()), "r"))
        return tp + 0.4 * tr, tp, tr
    lib.creid_tune_clear()
    tuned_keys.add((0, (M, cout, K, s << 1)))
    base, bp, br = wgrad_score()
    best = (base, None)
    cands = []
    for tm, tn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        if cout % tm or K % tn or cin < tn:
            continue
        tiles = (cout // tm) * (K // tn)
        # (round 5: besides the fixed ladder, the split counts that FILL the workgroup slots of the chip exactly -- 256 CUs x 1 ... 4
        #  resident workgroups --: a grid a few workgroups above a full round runs a second round on an almost empty chip, tools/grid_tail.py)
        fill = sorted({slots // tiles for slots in (256, 512, 768, 1024) if slots // tiles >= 1} - set(SPLITS))
        for sp in SPLITS + fill:
            if not (160 <= tiles * sp <= 1600) or sp > M // 64:
                continue
            lib.creid_tune_set(0, M, cout, K, s << 1, tm, tn, sp)
            sc, _, _ = wgrad_score()
            cands.append((sc, (tm, tn, sp)))
            if sc < best[0]:
                best = (sc, (tm, tn, sp))
    # two k-groups per workgroup (512 threads, plan word bit 21): half the workgroups for the same wave count, half the partials
    for tm, tn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        if cout % tm or K % tn or cin < tn:
            continue
        tiles = (cout // tm) * (K // tn)
        fill = sorted({slots // tiles for slots in (256, 512) if slots // tiles >= 1} - set(SPLITS))
        for sp in SPLITS + fill:
            if not (96 <= tiles * sp <= 800) or sp > M // 128:
                continue
            for depth in (0, 3):
                if depth == 3 and 2 * 3 * (tm + tn) * 128 > 160 * 1024:
                    continue
                word = sp | (1 << 21) | (depth << 16)
                lib.creid_tune_set(0, M, cout, K, s << 1, tm, tn, word)
                sc, _, _ = wgrad_score()
                if sc < best[0]:
                    best = (sc, (tm, tn, word))
    # ring depth 3 / producer-consumer split on the three best (tile, split) choices (plan word: splits | depth<<16 | ws<<20)
    for sc0, (tm, tn, sp) in sorted(cands)[:3]:
        for extra in ((3 << 16), (1 << 20)):
            if (extra >> 16) == 3 and 3 * (tm + tn) * 128 > 160 * 1024:
                continue
            lib.creid_tune_set(0, M, cout, K, s << 1, tm, tn, sp | extra)
            sc, _, _ = wgrad_score()
            if sc < best[0]:
                best = (sc, (tm, tn, sp | extra))
    lib.creid_tune_clear()
    if best[1] is not None and best[0] < 0.97 * base:
        plans.append({"kind": 0, "key": [M, cout, K, s << 1], "plan": list(best[1]), "us": round(best[0], 2), "rule_us": round(base, 2), "layer": name})
    bw = best[1][2] if best[1] else 0
    log.append(f"wgrad {name:28s} x{cnt} rule {base:6.1f} (partials {bp:5.1f} + reduce {br:5.1f})  best {best[0]:6.1f} "
               f"{best[1][:2] if best[1] else None} splits {bw & 0xffff} depth {(bw >> 16) & 15} ws {(bw >> 20) & 1} kg2 {(bw >> 21) & 1}")
    if args.wgrad_only:
        print(log[-1], flush=True)
        continue

    # ---------------- forward and data gradient
    # the data gradient as the training schedule runs it: with the next BatchNorm-backward's column sums in the epilogue
    # (operand x, ReLU bits, statistics) and, for the block's first convolution (cin = 4 x cout), the masked residual add
    Min = B * h * w
    bn_x = torch.randn((Min, cin), device="cuda").to(torch.bfloat16)
    bn_mask = torch.randint(0, 256, (Min * cin // 8,), dtype=torch.uint8, device="cuda")
    bn_mean, bn_inv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
    bn_part = torch.empty((lib.creid_bn2d_bwd_rows(Min) * 2, cin), device="cuda")
    add = torch.randn((Min, cin), device="cuda").to(torch.bfloat16) if (cin == 4 * cout and s == 1) else None
    dxbuf = torch.empty((Min, cin), device="cuda", dtype=torch.bfloat16)

    def dgrad_fused():
        L.check(lib.creid_conv2d_dgrad_fused_nhwc(C.byref(d), L.ptr(y), L.ptr(crsk), L.ptr(dxbuf), L.ptr(add), 1,
                                                  L.ptr(bn_mask) if add is not None else None, L.ptr(bn_x), None, L.ptr(bn_mask),
                                                  L.ptr(bn_mean), L.ptr(bn_inv), L.ptr(bn_part), 0, None, None, 0, None, 0, L.BF16,
                                                  L.stream()), "dgrad_fused")

    for tag, fn, key in (("fwd", lambda: ly.conv2d_fwd(x, krsc, s, pad, with_stats=True), (M, cout, K, s << 1)),
                         ("dgrad", dgrad_fused, (B * h * w, cin, k * k * cout, 1 | (s << 1)))):
        lib.creid_tune_clear()
        tuned_keys.add((1, tuple(key)))
        base = t_us(fn)
        best = (base, None)
        for bn in (64, 128):
            if key[1] % bn:
                continue
            for st in (2, 3, 4):
                if bn == 128 and st == 4:
                    continue
                lib.creid_tune_set(1, *key, bn, st, 0)
                sc = t_us(fn)
                if sc < best[0]:
                    best = (sc, (bn, st, 0))
                if tag == "fwd" and st <= 3:               # the 4-wave kernel (all waves issue DMA and multiply): forward only
                    lib.creid_tune_set(1, *key, bn, st, 1)
                    sc = t_us(fn)
                    if sc < best[0]:
                        best = (sc, (bn, st, 1))
        if tag == "fwd" and k == 1 and s == 1 and cin in (64, 128, 256):   # the persistent weights-in-LDS kernel (conv_stream.hip)
            lib.creid_tune_set(1, *key, 64, 2, 2)
            sc = t_us(fn)
            if sc < best[0]:
                best = (sc, (64, 2, 2))
        if tag == "fwd" and k == 1 and s == 1 and cin in (64, 128, 256):    # its second form; the ring-depth slot = column-slab cap
            for st in (2, 3, 4):
                lib.creid_tune_set(1, *key, 128, st, 4)
                sc = t_us(fn)
                if sc < best[0]:
                    best = (sc, (128, st, 4))
        lib.creid_tune_clear()
        if best[1] is not None and best[0] < 0.97 * base:
            plans.append({"kind": 1, "key": list(key), "plan": list(best[1]), "us": round(best[0], 2), "rule_us": round(base, 2), "layer": f"{tag} {name}"})
        log.append(f"{tag:5s} {name:28s} x{cnt} rule {base:6.1f}  best {best[0]:6.1f} {best[1]}")
    print(log[-3]); print(log[-2]); print(log[-1], flush=True)

if args.merge and os.path.exists(args.merge):
    old = json.load(open(args.merge)).get("plans", [])
    # entries of shapes measured in this run are replaced (or dropped, when the built-in rule now wins); others are kept
    if args.pp_only:
        # only kind-5 entries are owned by this mode: an older plan of the same key stays unless a kind-5 plan now beats it
        newkeys = {(e["kind"], tuple(e["key"])) for e in plans}
        plans = [e for e in old if (e["kind"], tuple(e["key"])) not in newkeys and not (e["plan"][2] == 5 and (e["kind"], tuple(e["key"])) in tuned_keys)] + plans
    else:
        plans = [e for e in old if (e["kind"], tuple(e["key"])) not in tuned_keys] + plans
out = {"_comment": "measured launch plans (tools/tune_plans.py) for the ResNet50 layer mix on one MI355X (B=64 256x128 = BASELINE "
                   "configs[1]; B=56 320x320 = configs[3] training; B=128 256x128 and B=256 320x320 forward = the eval-mode embedding batches); kind 0 = weight gradient (M, out_c, K) -> (tile rows, tile cols, splits), "
                   "kind 1 = forward / data gradient (M, N, K, transposed) -> (N tile, ring depth, kernel: 0 producer/consumer, 1 four-wave DMA, 2 persistent 1x1, 3 256-row tiles, 4 persistent 1x1 second form with the ring-depth slot as column-slab cap, 5 all-waves-multiply persistent kernel of conv_pipe.hip with plan[0] = its variant word; key[3] bit 3 = measured with the folded eval-mode epilogue).  Shapes without an entry use the "
                   "built-in rules.",
       "device": torch.cuda.get_device_name(0), "plans": plans}
json.dump(out, open(args.out, "w"), indent=1)
print(f"{len(plans)} plans written to {args.out}")
