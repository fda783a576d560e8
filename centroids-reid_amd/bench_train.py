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

PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")   # newest committed counter passes first
# hipGraph captures run in THREAD-LOCAL capture mode: in the default (global) mode ANY thread's event query is illegal while a
# capture is open, and the NCCL/RCCL watchdog thread polls the events of earlier collectives (the eager warm-up steps, the
# weight broadcast) -- found by `CREID_FORCE_DIST=1 python bench.py` on one GPU: "operation not permitted when stream is
# capturing" in ProcessGroupNCCL's watchdog, process aborted.  Every data-parallel run would have died at its first capture.
CAPTURE_MODE = "thread_local"
R50_FWD_BWD_GFLOP_PER_IMG = 24.32     # BASELINE.md section 3 (3 x forward conv FLOPs)
MFMA_BF16_TFLOPS = 2500.0
MFMA_F32_TFLOPS = 157.3


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


HBM_ACHIEVABLE_GBS = 6300.0      # MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy (8.0 TB/s spec)


def conv_launch_work(B, shape, what, role=None, first_block=False):
    """(FLOPs, algorithmic HBM bytes) of ONE convolution launch of the training step; shape = (cin, cout, k, stride, hin, win).
    Bytes = the activation tensors the launch must read and write once (bf16) plus its weights.  The data gradients carry
    fused passes whose operands count: c3 / c2: the column sums of the next BatchNorm backward (that layer's raw output + its
    ReLU bits); c1: the block's incoming gradient added through the ReLU bits and, except in the first block, the column sums
    of the previous block's bn3.  Weight gradient: both activation operands + the fp32 gradient it produces."""
    cin, cout, k, st, h, w = shape
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    fl = 2.0 * B * ho * wo * cout * cin * k * k
    x, y, wb = B * h * w * cin * 2, B * ho * wo * cout * 2, cout * cin * k * k * 2
    if what == "wgrad":
        return fl, x + y + 2 * wb                      # fp32 gradient = 2 x the 16-bit weight bytes
    if what == "dgrad" and st == 2 and k == 1:
        x = B * ho * wo * cin * 2                      # computed compact on the output grid, scatter-added by the c1 data gradient
    extra = 0
    if what == "dgrad" and role in ("c3", "c2"):
        extra = x + x // 16
    elif what == "dgrad" and role == "c1":
        extra = x + x // 16
        if not first_block:
            extra += x + x // 16
    return fl, x + y + wb + extra


def conv_step_sol(B, H, W, peak_tflops=MFMA_BF16_TFLOPS, hbm_gbs=HBM_ACHIEVABLE_GBS):
    """Speed of light of the step's 158 convolution launches (53 forward, 52 data gradients, 53 weight gradients), launch by
    launch: max(2 M N K / dense MFMA peak, algorithmic bytes / achievable HBM rate), summed per pass.  The attainable bound the
    measured in-situ time of the two convolution families is to be read against."""
    out = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    sh, i, bi = conv_shapes(B, H, W), 0, 0

    def add(what, shape, role, first):
        fl, by = conv_launch_work(B, shape, what, role, first)
        out[what] += max(fl / (peak_tflops * 1e12), by / (hbm_gbs * 1e9)) * 1e6
    stem_fl = 2.0 * B * (H // 2) * (W // 2) * 64 * 147
    stem_by = B * (H + 8) * (W + 6) * 4 * 2 + B * (H // 2) * (W // 2) * 64 * 2
    out["fwd"] += max(stem_fl / (peak_tflops * 1e12), stem_by / (hbm_gbs * 1e9)) * 1e6
    out["wgrad"] += max(stem_fl / (peak_tflops * 1e12), stem_by / (hbm_gbs * 1e9)) * 1e6
    for _planes, n in zip((64, 128, 256, 512), (3, 4, 6, 3)):
        for b in range(n):
            roles = [("c1", sh[i]), ("c2", sh[i + 1]), ("c3", sh[i + 2])]
            i += 3
            if b == 0:
                roles.append(("ds", sh[i])); i += 1
            for role, shape in roles:
                add("fwd", shape, role, bi == 0)
                add("dgrad", shape, role, bi == 0)
                add("wgrad", shape, role, bi == 0)
            bi += 1
    out["total"] = out["fwd"] + out["dgrad"] + out["wgrad"]
    return out


def embed_hbm_bytes(B, H, W):
    """Algorithmic HBM bytes of ONE eval-mode embedding forward (BatchNorm folded): every convolution reads its input and writes
    its output once (16-bit), conv3 also reads the block's residual, plus the weights, the padded image and the fused stem's
    pooled output.  ~40 MB per image at 256 x 128: the forward is HBM-bound (8.11 GFLOP / image = 3.2 us at the MFMA peak,
    40 MB = 6.4 us at 6.3 TB/s)."""
    by = B * (H + 8) * (W + 6) * 4 * 2 + B * (H // 4) * (W // 4) * 64 * 2
    sh, i = conv_shapes(B, H, W), 0
    for _planes, n in zip((64, 128, 256, 512), (3, 4, 6, 3)):
        for b in range(n):
            for j in range(4 if b == 0 else 3):
                cin, cout, k, st, h, w = sh[i + j]
                ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
                by += B * h * w * cin * 2 + B * ho * wo * cout * 2 + cout * cin * k * k * 2
                if j == 2:
                    by += B * ho * wo * cout * 2          # the residual (block input or the downsample branch's output)
            i += 4 if b == 0 else 3
    return by


def igemm_step_flops(B, H, W):
    """Algorithmic FLOPs of every launch of the igemm family in one training step: forward + data gradient of the 52
    non-stem convolutions (2 * M * N * K each) and the stem's forward (7x7x3 taps)."""
    fl = 0.0
    for cin, cout, k, s, h, w in conv_shapes(B, H, W):
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        fl += 2 * (2.0 * B * ho * wo * cout * cin * k * k)
    fl += 2.0 * B * (H // 2) * (W // 2) * 64 * 147
    return fl


def forward_flops(B, H, W, arch="resnet50"):
    """Forward-only convolution FLOPs (8.11 GFLOP / image R50 256x128, 25.33 IBN-a 320x320 -- same formula)."""
    fl = 2.0 * B * (H // 2) * (W // 2) * 64 * 147
    for cin, cout, k, s, h, w in conv_shapes(B, H, W):
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        fl += 2.0 * B * ho * wo * cout * cin * k * k
    return fl


_GROUPS = [("igemm", r"igemm|conv3x3_c64|stem_pool|c3_c1_kernel"), ("wgrad", r"wgrad_bf16|wgrad_f32|wgrad_reduce"), ("bn", r"bn2d_|ibn_|bn_apply|bn_bwd|bn_fold|col_stats"),
           ("heads", r"heads_stage|triplet|center_|xent|bn1d|loo_|gemm_f32|ctl_step|scale_matrix"), ("optim", r"adam|sgd_scaled|amp_"),
           ("pool_layout", r"maxpool|gap_|weight_prep|image_pad|nhwc")]


def parse_step_trace(db_path):
    """rocprofv3 kernel-trace .db of `bench.py --inner-trace` -> per-step kernel time by family, separately for the captured
    training step (segments that contain the Adam kernel) and the eval-mode embedding forward (segments that do not).
    A segment = the kernels from one image_pad launch to the next; the last segment of each kind is dropped (it runs
    into whatever follows) and up to three of the remaining ones are averaged."""
    import re
    import sqlite3
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    segs = {"train": [], "embed": []}
    for a, b in zip(marks, marks[1:] + [len(rows)]):
        seg = rows[a:b]
        segs["train" if any("adam" in r[0] for r in seg) else "embed"].append(seg)
    out = {}
    for kind, ss in segs.items():
        ss = ss[:-1][-3:]
        if not ss:
            continue
        acc = {}
        for seg in ss:
            for n, st, en in seg:
                g = next((gn for gn, pat in _GROUPS if re.search(pat, n)), "other")
                v = acc.setdefault(g, [0, 0.0]); v[0] += 1; v[1] += (en - st) / 1e3
            v = acc.setdefault("_span", [0, 0.0]); v[0] += 1; v[1] += (seg[-1][2] - seg[0][1]) / 1e3
        out[kind] = {g: {"launches": c / len(ss), "us": t / len(ss)} for g, (c, t) in acc.items()}
        out[kind]["_kernels"] = sum(v["launches"] for g, v in out[kind].items() if not g.startswith("_"))
        out[kind]["_kernel_us"] = sum(v["us"] for g, v in out[kind].items() if isinstance(v, dict) and not g.startswith("_"))
    return out


def insitu_trace(timeout_s=180):
    """IN-SITU kernel durations: re-runs this benchmark's captured training step and embedding forward for a few replays
    under `rocprofv3 --kernel-trace` in a child process and reads the per-launch durations back (parse_step_trace).  The
    per-family times are those of the launches inside the replayed step -- cold / warm operands, piggy-backed reductions
    and fused epilogues included -- which is what the step pays, unlike the isolated per-shape timings.  Returns None
    when rocprofv3 is unavailable or the child fails (the caller then reports the isolated figure only)."""
    import glob
    import shutil
    import subprocess
    import sys
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or os.environ.get("CREID_BENCH_NO_INSITU", "0") == "1":
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="creid_insitu_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", CREID_BENCH_NO_EVAL="1", CREID_BENCH_NO_INSITU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CREID_FORCE_DIST"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--stats", "-d", tmp, "-o", "insitu", "--", sys.executable, os.path.join(root, "bench.py"),
           "--inner-trace"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            print(f"[bench] in-situ trace failed (rc {r.returncode}): {r.stderr[-400:]}", file=sys.stderr)
            return None
        res = parse_step_trace(dbs[0])
        keep = os.path.join(root, "gpurun_out")
        if os.path.isdir(keep):
            import json
            with open(os.path.join(keep, "insitu_step_trace.json"), "w") as f:
                json.dump(res, f, indent=1)
        return res
    except Exception as e:  # noqa: BLE001  (a profiler problem must not cost the benchmark line)
        print(f"[bench] in-situ trace unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


_IGEMM_PAT = r"igemm_bf16_|igemm_bf16_pp|igemm1x1_stream|conv3x3_c64"
_WGRAD_PAT = r"wgrad_bf16_|wgrad_f32"


def pmc_insitu(timeout_s=150):
    """HBM-side traffic and matrix-pipe occupancy of the two convolution families INSIDE the replayed training step, measured in
    this run: three `rocprofv3 --kernel-trace --pmc` child runs of `bench.py --inner-trace` (FETCH_SIZE, WRITE_SIZE and
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE each in a pass of their own, as MI355X_MICROARCH.md prescribes; counters only, no
    other trace domain).  Per family and per STEP: launches, fetch_bytes = 2 x FETCH_SIZE (the gfx950 correction: the counter
    tallies 128-byte requests at 64), write_bytes = WRITE_SIZE, both KiB -> bytes; mfma_busy = busy cycles / (GRBM_GUI_ACTIVE / 8
    XCDs x 1024 SIMDs).  None when rocprofv3 or a pass fails (the line then says so instead of quoting a committed file)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import sys
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or os.environ.get("CREID_BENCH_NO_PMC", "0") == "1":
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR="/tmp", CREID_BENCH_NO_EVAL="1", CREID_BENCH_NO_INSITU="1", CREID_INNER_TRAIN_ONLY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CREID_FORCE_DIST"):
        env.pop(k, None)

    def one_pass(counters):
        tmp = tempfile.mkdtemp(prefix="creid_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "-d", tmp, "-o", "pmc", "--", sys.executable,
                   os.path.join(root, "bench.py"), "--inner-trace"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                print(f"[bench] counter pass {counters} failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, counter_name, value, dispatch_id, start from counters_collection").fetchall()
            disp = {}
            for kn, cn, v, did, st in rows:
                d = disp.setdefault(did, {"name": kn, "start": st})
                d[cn] = d.get(cn, 0.0) + float(v)
            order = sorted(disp.values(), key=lambda d: d["start"])
            marks = [i for i, d in enumerate(order) if "image_pad" in d["name"]]
            segs = [order[a:b] for a, b in zip(marks, marks[1:] + [len(order)])]
            segs = [sg for sg in segs if any("adam" in d["name"] for d in sg)]
            return segs[:-1][-2:] or None                  # complete replayed steps (the last segment runs into the teardown)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] counter pass {counters} unavailable ({type(e).__name__}: {e})", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def fam_sum(segs, pat, counter):
        tot = n = 0.0
        for sg in segs:
            for d in sg:
                if re.search(pat, d["name"]) and counter in d:
                    tot += d[counter]; n += 1
        return tot / len(segs), n / len(segs)

    F = one_pass(["FETCH_SIZE"])
    Wr = one_pass(["WRITE_SIZE"]) if F else None
    S = one_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]) if Wr else None
    if not (F and Wr):
        os.environ["CREID_BENCH_PMC_FAILED"] = "1"          # the evaluation's counter passes (bench.pmc_eval_insitu) are not tried either
        return None
    out = {}
    for key, pat in (("igemm", _IGEMM_PAT), ("wgrad", _WGRAD_PAT)):
        f, n = fam_sum(F, pat, "FETCH_SIZE")
        w, _ = fam_sum(Wr, pat, "WRITE_SIZE")
        e = {"launches": n, "fetch_bytes": 2.0 * f * 1024.0, "write_bytes": w * 1024.0}
        if S:
            b, _ = fam_sum(S, pat, "SQ_VALU_MFMA_BUSY_CYCLES")
            g, _ = fam_sum(S, pat, "GRBM_GUI_ACTIVE")
            e["mfma_busy"] = b / (g / 8.0 * 1024.0) if g else None
        out[key] = e
    return out


class EmbedBench:
    """validation_step's device work (modelling/bases.py:169-177): eval-mode backbone -> GAP -> BNNeck on a resident batch,
    captured once into a hipGraph."""

    def __init__(self, arch="resnet50", B=128, H=256, W=128, dtype=torch.bfloat16):
        self.arch, self.B, self.H, self.W, self.dtype = arch, B, H, W, dtype
        self.model = make_model(arch=arch, dtype=dtype)
        self.model.eval()
        gen = torch.Generator(device="cuda").manual_seed(5)
        self.x = torch.randn((B, 3, H, W), generator=gen, device="cuda", dtype=torch.float32)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._fwd()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            self.emb = self._fwd()

    def _fwd(self):
        with torch.no_grad():
            _, f = self.model.backbone(self.x)
            return self.model.bn(f)

    def run(self, steps, warmup):
        for _ in range(warmup):
            self.graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps


def run_embed(arch="resnet50", B=128, H=256, W=128, steps=20, warmup=3, insitu=None, label=None, dtype=torch.bfloat16):
    """The `embed` object of the bench line: images/s of the eval-mode embedding forward (BatchNorm folded into the
    convolution epilogues), its roofline against the bf16 / f16 MFMA peak (the same 2.5 PF)."""
    eb = EmbedBench(arch, B, H, W, dtype=dtype)
    dt = eb.run(steps, warmup)
    assert bool(torch.isfinite(eb.emb).all()), "non-finite embeddings in the benchmark"
    fl = forward_flops(B, H, W)                                # convolution FLOPs are the same for both backbones
    peak_tf = MFMA_F32_TFLOPS if dtype == torch.float32 else MFMA_BF16_TFLOPS
    res = {"metric": "embed_images_per_sec", "value": B / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps,
           "dtype": {torch.float16: "f16", torch.float32: "f32"}.get(dtype, "bf16"), "hip_graph": True,
           "config": {"workload": label or f"{arch} {H}x{W} eval-mode embedding forward (validation_step: backbone + GAP + BNNeck), "
                                           f"batch {B}, BatchNorm folded into the conv epilogues", "batch": B},
           "roofline": {"kernel": "convolution kernels of the forward (53 convolutions in 51 launches: igemm_bf16_{ws,dma,pp}, igemm1x1_stream2, conv3x3_c64, stem_pool, c3_c1; folded BN epilogue)", "bound": "mfma",
                        "achieved": fl / dt / 1e12, "peak": peak_tf, "unit": "TFLOP/s", "frac": fl / dt / 1e12 / peak_tf,
                        "source": "whole forward (wall time of the replayed graph, all kernels)", "traffic": None}}
    if arch == "resnet50":
        # the forward is HBM-bound (bytes / image model: embed_hbm_bytes): the honest yardstick of the whole forward
        by = embed_hbm_bytes(B, H, W)
        es = 4 if dtype == torch.float32 else 2
        by = by * es // 2
        res["roofline_hbm"] = {"bound": "hbm", "achieved": by / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by / dt / 1e9 / 8000.0,
                               "frac_of_achievable": by / dt / 1e9 / HBM_ACHIEVABLE_GBS, "algorithmic_bytes_per_image": by / B,
                               "sol_us_per_image": max(by / B / (HBM_ACHIEVABLE_GBS * 1e9), fl / B / (MFMA_BF16_TFLOPS * 1e12)) * 1e6,
                               "us_per_image": dt / B * 1e6, "traffic": None,
                               "source": "whole forward (wall time of the replayed graph) against the algorithmic activation + weight bytes"}
    if insitu and "embed" in insitu and "igemm" in insitu["embed"]:
        e = insitu["embed"]
        ig = e["igemm"]["us"] * 1e-6
        res["roofline"].update({"achieved": fl / ig / 1e12, "frac": fl / ig / 1e12 / peak_tf,
                                "frac_whole_forward": fl / dt / 1e12 / peak_tf,
                                "source": "rocprofv3 kernel trace of this run's replayed forward (in situ)",
                                "igemm_us": e["igemm"]["us"], "launches_per_forward": e["_kernels"]})
    del eb
    torch.cuda.empty_cache()
    return res


def embedding_errors(B=32, H=256, W=128):
    """Max-abs / relative error of the bf16 and f16 eval-mode embeddings (BNNeck output, L2-normalised like R1_mAP does) against
    the exact-f32 mode on the same weights and images -- the number to read next to every 16-bit throughput figure (north_star:
    fp32 embeddings within 1e-4; the 16-bit modes are throughput modes and do NOT meet it)."""
    out, ref = {}, None
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((B, 3, H, W), generator=gen, device="cuda", dtype=torch.float32)
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        torch.manual_seed(3)
        m = make_model(dtype=dt)
        m.eval()
        with torch.no_grad():
            _, f = m.backbone(x)
            e = torch.nn.functional.normalize(m.bn(f).float(), dim=1)
        if ref is None:
            ref = e
        else:
            d = (e - ref).abs()
            out[name] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "min_cosine": float((e * ref).sum(1).min())}
        del m
    torch.cuda.empty_cache()
    out["note"] = f"{B} synthetic images, random-init weights (seed 3), unit-length embeddings; fp32 mode = reference parity mode"
    return out


def run_embed_ranks(world, barrier_sync, B=128, H=256, W=128, steps=20, warmup=3):
    """Embeddings/s of the whole job (north_star's multi-GPU target is quoted in embeddings/s): every rank runs the eval-mode
    embedding forward on its own resident batch -- the partitioning of validation_step, no collective on the data path --
    between two barriers; the slowest rank's time counts.  Called by EVERY rank."""
    eb = EmbedBench("resnet50", B, H, W)
    for _ in range(warmup):
        eb.graph.replay()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        eb.graph.replay()
    barrier_sync(world)
    tmax = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item()) / steps
    ok = bool(torch.isfinite(eb.emb).all())
    del eb
    torch.cuda.empty_cache()
    assert ok, "non-finite embeddings in the benchmark"
    return {"metric": "embed_images_per_sec", "value": world * B / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps,
            "dtype": "bf16", "hip_graph": True, "scaling": "weak",
            "config": {"workload": f"resnet50 {H}x{W} eval-mode embedding forward (validation_step: backbone + GAP + BNNeck), batch {B} "
                                   "per rank, BatchNorm folded into the conv epilogues; max over ranks between two barriers",
                       "batch_per_rank": B, "parallelism": f"dp{world}"}}


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
    hyper = torch.tensor([3.5e-4, 1.0, 0.1, 0.03, 0.0, 0.0, 0.0, 0.0], device=dev)      # float[8]: {lr, step, bc1, bc2s, ticket, reserved}
    t_adam = time_kernel(lambda: L.check(lib.creid_adam_step_dev(L.ptr(p), L.ptr(gr), L.ptr(m1), L.ptr(v1), n, L.ptr(hyper),
                                                                    0.9, 0.999, 1e-8, 5e-4, 1.0, L.stream()), "adam"), 10)
    # input transforms (flip, pad-crop, ToTensor, Normalize, RandomErasing) of a uint8 batch, written straight into the stem's
    # padded NHWC4 operand: 3 bytes in, 8 out per pixel (+ the zero border)
    from .transforms import DeviceTransform
    tr = DeviceTransform((H, W), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], is_train=True, padding=10)
    Ba = 4 * B                                                   # a batch this small is a 10 us kernel: time four of them at once
    u8 = torch.randint(0, 256, (Ba, H, W, 3), dtype=torch.uint8, device=dev)
    prm = torch.from_numpy(tr.draw(Ba)).to(dev)
    t_aug = time_kernel(lambda: tr(u8, prm, layout="stem", dtype=torch.bfloat16), 20)
    aug_bytes = Ba * H * W * 3 + Ba * (H + 8) * (W + 6) * 4 * 2
    e = M * Cc * 2
    return {"augment_u8_GBs": aug_bytes / (t_aug * 1e-3) / 1e9, "augment_u8_us_per_batch": t_aug * 1e3 / 4,
            "bn2d_apply_GBs": 3 * e / (t_apply * 1e-3) / 1e9,                 # read x, residual; write y
            "bn2d_bwd_GBs": (3 * e + 3 * e + 2 * e) / (t_bwd * 1e-3) / 1e9,     # reduce: x,g,act; apply: x,g,act -> dx,gm
            "adam_GBs": 7 * n * 4 / (t_adam * 1e-3) / 1e9,                     # read p,g,m,v; write p,m,v
            "peak_GBs": 8000.0}


def pmc_field(key, field):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        for name in PMC_FILES:
            path = os.path.join(root, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)[key][field]
    except (OSError, KeyError, ValueError):
        pass
    return None


def pmc_traffic(key):
    """Average HBM-side bytes per launch of a kernel family from the committed rocprofv3 --pmc passes
    (profiles/r0x_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE -- FETCH_SIZE reads 0.5x on 16-byte streaming loads, see
    the calibration block of that file -- collected in their own counter-only runs); None if absent."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in PMC_FILES:
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
    for dt in (torch.float32, torch.bfloat16, torch.float16):
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
    for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16"), (torch.float16, "f16")):
        cmc, mAP, _ = rm.R1_mAP(num_query=len(q_rows), streamed=True).compute(embs[dt], pids, cams)
        res[name] = (mAP, float(cmc[0]))
    a, b, h = embs[torch.float32], embs[torch.bfloat16], embs[torch.float16]
    cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
    cos_h = torch.nn.functional.cosine_similarity(a, h, dim=1)
    return {"mAP_f32": res["f32"][0], "mAP_bf16": res["bf16"][0], "mAP_bf16_minus_f32": res["bf16"][0] - res["f32"][0],
            "rank1_f32": res["f32"][1], "rank1_bf16": res["bf16"][1], "min_cosine": float(cos.min().item()),
            # the reference's own mixed precision (utils/misc.py:111, precision=16) as the compute type of the eval forward
            "mAP_f16": res["f16"][0], "mAP_f16_minus_f32": res["f16"][0] - res["f32"][0], "rank1_f16": res["f16"][1],
            "min_cosine_f16": float(cos_h.min().item()),
            "images": int(len(x)), "identities": n_id, "noise": noise}


def train_curve_delta(P=16, K=4, H=256, W=128, steps=50, n_id=64, noise=0.6):
    """bf16 throughput mode against the exact-f32 parity mode as a TRAINING TRAJECTORY: the same randomly initialised model (same
    seed), the same `steps` PK batches of clustered synthetic identities (image = smooth per-identity pattern + N(0, noise)
    pixels; 64 identities, every one revisited each 4 steps), the full step (four losses, backward, Adam + center SGD) -- loss
    per step in both modes.  Reported: the largest and the mean relative loss gap over the trajectory and both end points
    (`final_loss` of two runs of different length is not a comparison)."""
    gen = torch.Generator(device="cuda").manual_seed(1)
    base = torch.randn((n_id, 3, H // 16, W // 16), generator=gen, device="cuda")
    base = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    camid = torch.zeros(P * K, dtype=torch.int64)
    is_real = torch.ones(P * K, dtype=torch.bool)
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)                                        # identical initial weights in both modes
        model = make_model(num_classes=n_id, dtype=dt)
        g2 = torch.Generator(device="cuda").manual_seed(2)          # identical pixel noise in both modes
        losses = []
        for s in range(steps):
            ids = (np.arange(P) * 5 + s * P) % n_id                 # 16 distinct identities (5 is coprime to 64)
            x = base[torch.as_tensor(ids, device="cuda")].repeat_interleave(K, 0) \
                + noise * torch.randn((P * K, 3, H, W), generator=g2, device="cuda")
            labels = torch.as_tensor(np.repeat(ids, K).astype(np.int64), device="cuda")
            losses.append(model.training_step((x, labels, camid, is_real), s)["loss"].detach().float().reshape(()))
        curves[dt] = torch.stack(losses).cpu().numpy().astype(np.float64)
        del model
        torch.cuda.empty_cache()
    f, b = curves[torch.float32], curves[torch.bfloat16]
    rel = np.abs(b - f) / np.abs(f)
    return {"steps": steps, "identities": n_id, "noise": noise, "max_rel_loss_gap": float(rel.max()), "mean_rel_loss_gap": float(rel.mean()),
            "step_of_max": int(rel.argmax()), "loss_f32_first_last": [float(f[0]), float(f[-1])],
            "loss_bf16_first_last": [float(b[0]), float(b[-1])],
            "loss_f32_every_10th": [round(float(v), 4) for v in f[::10]], "loss_bf16_every_10th": [round(float(v), 4) for v in b[::10]],
            "note": "same seed, same batches, full CTL step; bf16 = MFMA inputs + stored activations, fp32 master weights / "
                    "accumulators / losses / optimiser"}


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
            for s in range(3):                              # eager warm-up (allocations, lazy inits), with real syncs; three
                self._eager_step(s)                         # steps like the single-graph path: same step count, comparable losses
            torch.cuda.synchronize()
            segs = [torch.cuda.CUDAGraph()]
            split_at = []

            def cap_hook(k):
                if k in (4, 3):
                    segs[-1].capture_end()
                    split_at.append(k)
                    g = torch.cuda.CUDAGraph()
                    segs.append(g)
                    g.capture_begin(pool=segs[0].pool(), capture_error_mode=CAPTURE_MODE)
            eng.on_group_done = cap_hook
            open_graph = None
            try:
                segs[0].capture_begin(capture_error_mode=CAPTURE_MODE)
                open_graph = segs
                self.out = model.forward_backward(self.static, 0)
                segs[-1].capture_end()
                open_graph = None
                eng.on_group_done = None
                self.gopt = torch.cuda.CUDAGraph()
                self.gopt.capture_begin(pool=segs[0].pool(), capture_error_mode=CAPTURE_MODE)
                open_graph = [self.gopt]
                model.apply_optimizers()
                self.gopt.capture_end()
                open_graph = None
            except BaseException:
                eng.on_group_done = None
                if open_graph is not None:               # never leave the stream in capture mode behind a failed capture
                    try:
                        open_graph[-1].capture_end()
                    except Exception:  # noqa: BLE001
                        pass
                raise
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


def inner_trace(args, barrier_sync):
    """`bench.py --inner-trace` (the child of insitu_trace): a few replays of the captured training step, then of the
    embedding forward, nothing else -- the parent reads the kernel durations from the profiler's trace."""
    args.steps, args.warmup = 5, 2
    run(args, 0, 1, barrier_sync, None, None, minimal=True)
    torch.cuda.synchronize()
    if os.environ.get("CREID_INNER_TRAIN_ONLY", "0") == "1":      # the counter passes of pmc_insitu look at the step only
        return
    eb = EmbedBench()
    eb.run(5, 2)
    torch.cuda.synchronize()


def fake_mix_step(model, batches, P, K, steps=30, warmup=5):
    """SURVEY 8d cfg2 variant: 10 % of the steps carry 1-2 padded (isReal = False) samples.  ONE captured graph serves every
    pattern: the step takes the mask from a device tensor (train_ctl_model._forward_backward_fused, masked schedule) and
    contains no host synchronisation; per step only the mask / image / label buffers are refreshed."""
    B = P * K
    dev = batches[0][0].device
    mask = torch.ones(B, dtype=torch.uint8, device=dev)
    sx, sl = batches[0][0].clone(), batches[0][1].clone()
    static = (sx, sl, batches[0][2], mask)
    pats = [torch.ones(B, dtype=torch.uint8, device=dev) for _ in range(10)]
    pats[3][5] = 0                                             # one fake in identity 1
    pats[3][K * 7 + 1] = 0; pats[3][K * 7 + 2] = 0             # two fakes in identity 7 (two real instances left)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in range(2):
            mask.copy_(pats[3 if s else 0])
            model.training_step(static, s)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
        out = model.training_step(static, 0)

    def one(s):
        mask.copy_(pats[s % 10]); sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
        graph.replay()
    for s in range(warmup):
        one(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        one(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(out["loss"])
    assert np.isfinite(loss), "non-finite loss in the fake-mix benchmark"
    return {"value": B / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps, "final_loss": loss,
            "note": "same step captured with a DEVICE isReal mask; every 10th step has 3 padded samples (1 + 2 in two identities)"}


def fp32_mode_step(P, K, H, W, steps=6, warmup=2, dtype=torch.float32):
    """The exact-f32 parity mode of the same training step (fp32 activations, v_mfma_f32_32x32x2_f32): the throughput that
    goes with the <= 1e-4 embedding / mAP parity claims (bf16 is the throughput mode).  dtype = torch.float16: the reference's own
    mixed precision (utils/misc.py:111) -- f16 MFMA inputs / activations / gradient tensors + the device-resident loss scale.
    Timed like the headline: a fresh synthetic batch is copied into the captured step's input buffers before every replay."""
    model = make_model(dtype=dtype)
    bs = [synthetic_batch(P, K, H, W, s) for s in range(4)]
    sx, sl = bs[0][0].clone(), bs[0][1].clone()
    static = (sx, sl, bs[0][2], bs[0][3])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in range(2):
            model.training_step(static, s)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
        out = model.training_step(static, 0)

    def one(s):
        sx.copy_(bs[s % 4][0]); sl.copy_(bs[s % 4][1])
        graph.replay()
    for s in range(warmup):
        one(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        one(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(out["loss"])
    del model, graph
    torch.cuda.empty_cache()
    if dtype == torch.float16:
        return {"value": P * K / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps, "dtype": "f16", "final_loss": loss,
                "note": "same step with f16 as the compute type (the reference's precision=16): loss scale 65536 resident on the "
                        "device, in-place unscale + non-finite check of the backbone gradients and the GradScaler update inside the "
                        "captured graph (5 more launches per step)"}
    return {"value": P * K / dt, "unit": "images/s", "ms_per_step": dt * 1e3, "steps": steps, "dtype": "f32",
            "final_loss": loss, "note": "same step, exact-f32 MFMA parity mode (the mode the <=1e-4 golden comparisons run in)"}


def run(args, rank, world, barrier_sync, time_kernel, cpu_baseline_fn=None, minimal=False):
    P, K, H, W = 16, 4, 256, 128
    arch = os.environ.get("CREID_BENCH_ARCH", "resnet50")      # resnet50_ibn_a: side measurement, not the headline config
    if os.environ.get("CREID_BENCH_CONFIG3", "0") == "1":      # side line: the training half of BASELINE configs[3]
        arch, P, H, W = "resnet50_ibn_a", 14, 320, 320         # (configs/320_resnet50_ibn_a.yml: 320 x 320, 14 x 4 images)
    if os.environ.get("CREID_BENCH_P"):                        # side measurement: another batch size (P identities x K = 4)
        P = int(os.environ["CREID_BENCH_P"])
    f32 = os.environ.get("CREID_BENCH_DTYPE", "bf16") == "f32"   # side measurement: the exact-f32 parity mode
    f16 = os.environ.get("CREID_BENCH_DTYPE", "bf16") == "f16"   # side measurement: f16 + device-resident loss scale (precision=16)
    torch.manual_seed(int(os.environ.get("CREID_BENCH_SEED", "0")))   # random-init weights: the same ones in every run
    model = make_model(arch=arch, dtype=torch.float32 if f32 else (torch.float16 if f16 else torch.bfloat16))
    ddp = world > 1 or (dist.is_available() and dist.is_initialized())      # CREID_FORCE_DIST=1: one-rank RCCL group
    overlap = False
    if ddp:
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
    split_graph = ddp or os.environ.get("CREID_SPLIT_GRAPH", "0") == "1"
    stepper = None
    if ddp and overlap:
        sx, sl = batches[0][0].clone(), batches[0][1].clone()
        import sys
        ok = 1
        try:
            stepper = DDPStepper(model, world, (sx, sl, batches[0][2], batches[0][3]), use_graph=use_graph)
        except Exception as e:  # noqa: BLE001  (capture problems must not cost the measurement: use the two-graph path)
            ok = 0
            print(f"[bench] rank {rank}: overlapped data-parallel step unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        # every rank must run the SAME collective schedule: agree on the outcome before choosing it (a rank that fell back
        # alone would issue one flat all-reduce against the others' three bucketed ones and the job would hang).  The eager
        # warm-up steps inside DDPStepper were data-parallel steps on every rank, so the replicas are still identical.
        flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if rank == 0:
                print("[bench] falling back to graph A -> all-reduce -> graph B on every rank", file=sys.stderr)
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
            with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
                gout = model.training_step(static, 0)

            def one_step(s):
                sx.copy_(batches[s % 4][0]); sl.copy_(batches[s % 4][1])
                graph.replay()
                return gout
        else:
            model.grad_sync = None
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode=CAPTURE_MODE):
                gout = model.forward_backward(static, 0)
            if sync is not None:
                sync(model)                      # also fixes opt.grad_scale = 1/world before graph B is captured
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode=CAPTURE_MODE):
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
    if minimal:
        return {"ms_per_step": dt / args.steps * 1e3}
    if rank == 0:
        loss = float(out["loss"])
        assert np.isfinite(loss), "non-finite loss in the benchmark"
        res["final_loss"] = loss
        if f16:
            opt0, _ = model.optimizers()
            res["f16_state"] = {"loss_scale": model.loss_scaler.get_scale(), "adam_steps_applied": opt0.step_count}
        res["host_enqueue_ms_per_step"] = t_host / args.steps * 1e3
        res["hip_graph"] = bool(use_graph)
        ms = dt / args.steps * 1e3
        gflop_img = 76.0 if (H, W) == (320, 320) else R50_FWD_BWD_GFLOP_PER_IMG      # IBN-a 320x320 / R50 256x128
        tf_step = gflop_img * P * K / (ms * 1e-3) / 1e3
        res["step_mfma_frac"] = tf_step / MFMA_BF16_TFLOPS     # all conv FLOPs / whole-step time (incl. HBM-bound BN etc.)
        from . import _lib as L
        res["plans"] = "tuned" if (L.lib() and L.N_PLANS > 0) else "rules"   # tuned_plans.json covers the configs[1] / [3] shapes only
        tf, ig_ms, slow, fast = igemm_roofline(P * K, H, W, time_kernel)
        headline = world == 1 and arch == "resnet50" and not f32 and not f16 and (H, W) == (256, 128) and P == 16
        insitu = insitu_trace() if headline else None
        pmc = pmc_insitu() if headline else None
        roof = {"kernel": "convolution forward + data-gradient kernels (igemm_bf16_{dma,ws,pp}, igemm1x1_stream2, conv3x3_c64; stem fwd; 105 launches/step, real layer mix)",
                "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_TFLOPS,
                "source": "HIP events, every shape launched alone back to back on warm operands (isolated)",
                "frac_isolated": tf / MFMA_BF16_TFLOPS, "isolated_ms_per_step": ig_ms,
                "traffic": None, "traffic_source": "not measured in this run (rocprofv3 counter passes unavailable or switched off)",
                "slowest_TFs": slow, "fastest_TFs": fast,
                # the whole step against the same peak: ALL convolution FLOPs (fwd + dgrad + wgrad) / the step's wall time -- what
                # north_star's ">= 70 % MFMA-roofline on ResNet50 fwd+bwd" is phrased in (the BatchNorm / optimiser passes are
                # HBM-bound and sit in the denominator)
                "frac_whole_step": tf_step / MFMA_BF16_TFLOPS}
        if pmc:
            e = pmc["igemm"]
            roof.update({"traffic": (e["fetch_bytes"] + e["write_bytes"]) / max(e["launches"], 1.0),
                         "traffic_source": "THIS run: rocprofv3 --kernel-trace --pmc child passes over the replayed step (FETCH_SIZE, "
                                           "WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, one pass each); bytes per launch = "
                                           "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches (gfx950: FETCH_SIZE tallies 128-byte requests at 64)",
                         "traffic_bytes_per_step": e["fetch_bytes"] + e["write_bytes"], "traffic_launches_per_step": e["launches"],
                         "mfma_busy_by_counter": e.get("mfma_busy")})
            res["wgrad_family_counters"] = {
                "launches_per_step": pmc["wgrad"]["launches"], "fetch_bytes_per_step": pmc["wgrad"]["fetch_bytes"],
                "write_bytes_per_step": pmc["wgrad"]["write_bytes"], "mfma_busy_by_counter": pmc["wgrad"].get("mfma_busy"),
                "algorithmic_bytes_per_step": sum(conv_launch_work(P * K, sh_, "wgrad")[1] for sh_ in conv_shapes(P * K, H, W)),
                "source": "same counter passes as roofline.traffic"}
            res["wgrad_family_counters"]["traffic_over_algorithmic"] = (
                (pmc["wgrad"]["fetch_bytes"] + pmc["wgrad"]["write_bytes"]) / res["wgrad_family_counters"]["algorithmic_bytes_per_step"])
        if insitu and "train" in insitu and "igemm" in insitu["train"]:
            t = insitu["train"]
            fl = igemm_step_flops(P * K, H, W)
            tf_in = fl / (t["igemm"]["us"] * 1e-6) / 1e12
            roof.update({"achieved": tf_in, "frac": tf_in / MFMA_BF16_TFLOPS,
                         "source": "rocprofv3 kernel trace of this run's replayed step: sum of 2*M*N*K over the family's launches / "
                                   "sum of their durations IN SITU (fused epilogues and piggy-backed reductions included)",
                         "algorithmic_TFLOP_per_step": fl / 1e12, "igemm_us_per_step": t["igemm"]["us"],
                         "igemm_launches_per_step": t["igemm"]["launches"]})
            res["step_anatomy_us"] = {g: round(v["us"], 1) for g, v in t.items() if isinstance(v, dict)}
            res["launches_per_step"] = t["_kernels"]
            # the attainable bound, launch by launch: sum over the 158 convolution launches of max(2MNK / 2.5 PF, algorithmic
            # bytes / 6.3 TB/s) against what those launches take inside the step (both convolution families, split reductions
            # included) -- `frac` above prices the kernels against the MFMA peak alone, this against what a perfect kernel could
            # reach at B = 64, where most layers are bandwidth- and latency-bound
            sol = conv_step_sol(P * K, H, W)
            conv_us = t["igemm"]["us"] + t.get("wgrad", {"us": 0.0})["us"]
            roof.update({"sol_us": sol["total"], "sol_us_by_pass": {k: round(v, 1) for k, v in sol.items() if k != "total"},
                         "measured_conv_us": conv_us, "frac_of_sol": sol["total"] / conv_us,
                         "sol_definition": "sum over the step's 158 convolution launches of max(2MNK / 2.5 PFLOP/s, algorithmic bytes / "
                                           "6.3 TB/s achievable HBM) -- bench_train.conv_step_sol; measured = in-situ time of the "
                                           "convolution forward / data-gradient / weight-gradient launches"})
        res["roofline"] = roof
        res["roofline_hbm_stages"] = hbm_stage_rates(time_kernel, P * K, H, W)
        if cpu_baseline_fn is not None and world == 1:          # the CPU leg is reported at N=1 only
            res["cpu_baseline"] = cpu_baseline_fn(P, K, H, W)
        if headline and os.environ.get("CREID_BENCH_NO_EVAL", "0") != "1":
            res["fake_mix"] = fake_mix_step(model, batches, P, K)
            res["fake_mix"]["vs_all_real"] = res["fake_mix"]["value"] / (imgs / dt)
            del model
            torch.cuda.empty_cache()
            res["fp32_mode"] = fp32_mode_step(P, K, H, W)
            res["f16_train"] = fp32_mode_step(P, K, H, W, steps=args.steps, warmup=args.warmup, dtype=torch.float16)
            res["f16_train"]["vs_bf16"] = res["f16_train"]["value"] / (imgs / dt)
            res["map_delta_bf16"] = map_delta_bf16()             # BASELINE metric (iii) on clustered synthetic identities
            md = res["map_delta_bf16"]
            res["map_delta_f16"] = {"mAP_f32": md["mAP_f32"], "mAP_f16": md["mAP_f16"], "mAP_f16_minus_f32": md["mAP_f16_minus_f32"],
                                    "rank1_f32": md["rank1_f32"], "rank1_f16": md["rank1_f16"], "min_cosine": md["min_cosine_f16"],
                                    "note": "same recipe as map_delta_bf16; f16 = compute type of the eval-mode forward (conv MFMA inputs "
                                            "and activations), fp32 accumulate"}
            # the same delta on two more recipes (easier / harder identities; at noise 0.3 every identity is separable and mAP is 1.0 in all three types): one synthetic point is one point
            res["map_delta_recipes"] = {}
            for nz in (0.5, 0.9):
                m2 = map_delta_bf16(noise=nz)
                res["map_delta_recipes"][f"noise_{nz}"] = {k: m2[k] for k in ("mAP_f32", "mAP_bf16_minus_f32", "mAP_f16_minus_f32", "rank1_f32",
                                                                              "rank1_bf16", "rank1_f16", "min_cosine", "min_cosine_f16")}
            res["train_curve_delta"] = train_curve_delta(P, K, H, W)
            # embeddings/s: the eval-mode forward validation_step / inference run (north_star's multi-GPU target is in it)
            res["embed"] = run_embed("resnet50", 128, 256, 128, steps=20, warmup=3, insitu=insitu)
            ef = run_embed("resnet50", 128, 256, 128, steps=20, warmup=3, dtype=torch.float16)
            res["embed"]["f16"] = {"value": ef["value"], "unit": ef["unit"], "ms_per_step": ef["ms_per_step"], "dtype": "f16",
                                   "vs_bf16": ef["value"] / res["embed"]["value"], "roofline_frac_whole_forward": ef["roofline"]["frac"]}
            # the accuracy-compliant mode (north_star: fp32 embeddings within 1e-4 of the reference): the same forward on the exact-f32
            # MFMA kernels, against the f32 matrix peak (157.3 TF/s); and what the 16-bit compute types cost in embedding error
            e32 = run_embed("resnet50", 128, 256, 128, steps=5, warmup=2, dtype=torch.float32)
            res["embed"]["fp32"] = {"value": e32["value"], "unit": e32["unit"], "ms_per_step": e32["ms_per_step"], "dtype": "f32",
                                    "vs_bf16": e32["value"] / res["embed"]["value"], "roofline": e32["roofline"],
                                    "roofline_hbm": e32.get("roofline_hbm"),
                                    "note": "the parity mode: embeddings <= 1e-4 of the reference's fp32 (tests/test_backbone_gpu.py goldens)"}
            res["embed"]["embedding_error_vs_fp32"] = embedding_errors()
            # the same forward at the batches inference.run_inference's macro-batching (and TEST.IMS_PER_BATCH 256 of the reference's
            # large configs) runs it at: from two tiles per persistent workgroup on the convolution kernels pipeline across tiles
            for bb_ in (256, 512):
                eb = run_embed("resnet50", bb_, 256, 128, steps=10, warmup=3)
                res["embed"][f"batch{bb_}"] = {"value": eb["value"], "unit": eb["unit"], "ms_per_step": eb["ms_per_step"], "batch": bb_,
                                                "vs_batch128": eb["value"] / res["embed"]["value"],
                                                "roofline_frac_whole_forward": eb["roofline"]["frac"]}
            res["embed"]["configs3_embedding_half"] = run_embed(
                "resnet50_ibn_a", 256, 320, 320, steps=5, warmup=2,
                label="ResNet50-IBN-a 320x320 eval-mode embedding forward, batch 256 (embedding half of BASELINE configs[3], "
                      "TEST.IMS_PER_BATCH 256)")
            e5 = run_embed("resnet50_ibn_a", 512, 320, 320, steps=4, warmup=2)
            res["embed"]["configs3_embedding_half"]["batch512"] = {
                "value": e5["value"], "unit": e5["unit"], "ms_per_step": e5["ms_per_step"], "batch": 512,
                "vs_batch256": e5["value"] / res["embed"]["configs3_embedding_half"]["value"],
                "note": "two loader batches per forward (inference.run_inference macro_batch)"}
    if ddp and arch == "resnet50" and not f32 and not f16 and (H, W) == (256, 128) and os.environ.get("CREID_BENCH_NO_EVAL", "0") != "1":
        er = run_embed_ranks(world, barrier_sync)            # every rank takes part; rank 0 reports
        if rank == 0:
            res["embed_ranks" if "embed" in res else "embed"] = er
    return {"metric": "train_images_per_sec", "value": imgs / dt, "unit": "images/s",
            "ms_per_step": dt / args.steps * 1e3, "dtype": "f32" if f32 else ("f16" if f16 else "bf16"),
            "config": {"workload": ("ResNet50" if arch == "resnet50" else arch) +
                                   f" {H}x{W} CTL training step: fwd+bwd, centroid-triplet + center + xent, "
                                   "Adam + center SGD (BASELINE configs[1])" if (H, W) == (256, 128) else
                                   f"{arch} {H}x{W} CTL training step, P={P} x K={K} (training half of BASELINE configs[3], side line)",
                       "P": P, "K": K, "global_batch": P * K * world, "num_classes": 751,
                       "parallelism": f"dp{world}" if world > 1 else "single"}, **res}
