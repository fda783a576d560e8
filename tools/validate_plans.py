"""In-situ validation of the kind-5 launch plans (conv_pipe.hip): a plan is measured by tools/tune_plans.py on ten back-to-back
launches of ONE layer with warm operands; inside the captured forward / step the same launch follows another kernel and meets
cold weights and an uneven tail of the previous grid -- several long-K plans that won by 5-14 % alone LOSE there
(profiles/r04_plan_validation.md).  This tool keeps a plan only if the whole captured workload is not faster without it.

    python tools/validate_plans.py [--configs embed128,embed256,train64,train56] [--plans centroids-reid_amd/tuned_plans.json]

Per configuration: time the workload with all plans, then once per kind-5 plan of that configuration with the plan removed
(hipGraph re-captured: the plan is read at launch time); a plan whose removal makes the workload faster by more than the noise
margin is dropped for good.  The embedding configurations run in-process (bench_train.EmbedBench), the training ones as
`bench.py` children (CREID_BENCH_NO_EVAL=1: the step only)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="embed128,embed256,train64,train56")
ap.add_argument("--plans", default=os.path.join(ROOT, "centroids-reid_amd", "tuned_plans.json"))
ap.add_argument("--margin", type=float, default=0.0015, help="relative gain of the whole workload that counts as real")
ap.add_argument("--all-kinds", action="store_true", help="validate EVERY plan of the configuration (weight-gradient and round-3 "
                                                          "forward / data-gradient plans too), not only kind 5")
ap.add_argument("--fast", action="store_true", help="one timing per trial instead of the minimum of several")
args = ap.parse_args()

CONFIGS = {  # name -> (kind, arch, B, H, W)
    "embed128": ("embed", "resnet50", 128, 256, 128),
    "embed256": ("embed", "resnet50_ibn_a", 256, 320, 320),
    "train64": ("train", "resnet50", 64, 256, 128),
    "train56": ("train", "resnet50_ibn_a", 56, 320, 320),
}
doc = json.load(open(args.plans))
plans = doc["plans"]


def conv_ms(B, H, W):
    """(M, N, K) of every forward convolution at this size (and of its data gradient) -> which plans belong to the configuration."""
    from centroids_reid_amd.bench_train import conv_shapes
    out = set()
    for cin, cout, k, s, h, w in conv_shapes(B, H, W):
        oh, ow = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        out.add((B * oh * ow, cout, cin * k * k))
        out.add((B * h * w, cin, cout * k * k))                  # data gradient
    return out


def write_tmp(entries):
    f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, dir="/tmp")
    json.dump({"plans": entries}, f)
    f.close()
    return f.name


def time_embed(entries, arch, B, H, W, dtype=None):
    import torch
    from centroids_reid_amd import _lib as L
    from centroids_reid_amd.bench_train import EmbedBench
    L.lib().creid_tune_clear()
    path = write_tmp(entries)
    L.load_tuned_plans(path)
    os.unlink(path)
    eb = EmbedBench(arch, B, H, W)
    t = min(eb.run(20, 3) for _ in range(1 if args.fast else 3))
    del eb
    torch.cuda.empty_cache()
    return t * 1e3


def time_train(entries, arch, B, H, W):
    path = write_tmp(entries)
    env = dict(os.environ, CREID_TUNED_PLANS=path, CREID_BENCH_NO_EVAL="1", CREID_BENCH_NO_INSITU="1")
    if (H, W) != (256, 128):
        env["CREID_BENCH_CONFIG3"] = "1"
    best = None
    for _ in range(1 if args.fast else 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
        best = ms if best is None else min(best, ms)
    os.unlink(path)
    return best


log = []
for name in args.configs.split(","):
    kind, arch, B, H, W = CONFIGS[name]
    mnk = conv_ms(B, H, W)
    bit = 8 if kind == "embed" else 0
    if args.all_kinds:
        # every plan a launch of this configuration can hit: eval launches fall back to mode-less keys, training ones never see bit 3
        mine = [e for e in plans if tuple(e["key"][:3]) in mnk and (kind == "train" or e["kind"] == 1) and
                (kind == "embed" or (e["key"][3] & 8) == 0)]
    else:
        mine = [e for e in plans if e["kind"] == 1 and e["plan"][2] == 5 and tuple(e["key"][:3]) in mnk and (e["key"][3] & 8) == bit]
    timer = (lambda ents: time_embed(ents, arch, B, H, W)) if kind == "embed" else (lambda ents: time_train(ents, arch, B, H, W))
    base = timer(plans)
    print(f"[{name}] {len(mine)} plans to validate, workload with all of them: {base:.4f} ms", flush=True)
    log.append(f"## {name}: {arch} {H}x{W} batch {B} ({'eval-mode embedding forward' if kind == 'embed' else 'training step'}); "
               f"all plans {base:.4f} ms\n\n| plan (M, N, K, mode) | variant | tuner: alone us (plan / before) | workload without it ms | verdict |\n|---|---|---|---:|---|")
    for e in mine:
        trial = [p for p in plans if p is not e]
        t = timer(trial)
        drop = t < base * (1.0 - args.margin)
        log.append(f"| kind {e['kind']} {tuple(e['key'])} | {e['plan']} | {e.get('us')} / {e.get('rule_us')} | {t:.4f} | "
                   f"{'DROPPED (workload %.2f %% faster without)' % (100 * (base - t) / base) if drop else 'kept'} |")
        print(log[-1], flush=True)
        if drop:
            plans, base = trial, t
    final = timer(plans)
    log.append(f"\nafter validation: {final:.4f} ms\n")
    print(f"[{name}] after validation {final:.4f} ms", flush=True)

doc["plans"] = plans
doc["_validated"] = "kind-5 plans validated in situ by tools/validate_plans.py (whole captured forward / step with and without each plan)"
json.dump(doc, open(args.plans, "w"), indent=1)
open(os.path.join(ROOT, "gpurun_out", "plan_validation_all.md" if args.all_kinds else "plan_validation.md"), "w").write("\n".join(log) + "\n")
print(f"{len(plans)} plans kept in {args.plans}")
