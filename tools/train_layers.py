"""Per-layer IN-SITU table of ONE replay of the captured training step: one row per convolution x
{fwd, dgrad, wgrad} from a rocprofv3 kernel-trace .db of `bench.py` (tools/prof_train.sh) -- layer shape, the kernel variant
that ran, its duration inside the step, achieved TFLOP/s (2 M N K) and the HBM rate of its algorithmic bytes (the activation
tensors it must read and write once, bf16).  Launches are labelled by POSITION: the engine's launch order is fixed
(centroids-reid_amd/backbone.py forward(): stem, then per block c1, c2, [ds], c3; backward(): per block in reverse the data
gradients c3, c2, [ds], c1 and the weight gradients c3, c2, [ds], c1; the stem's weight gradient last).
    python tools/train_layers.py <results.db> [batch H W]   -> profiles/r05_train_layers.md"""
import re
import sqlite3
import sys

sys.path.insert(0, ".")
from centroids_reid_amd.bench_train import conv_shapes, conv_launch_work, conv_step_sol   # noqa: E402

IS_CONV = re.compile(r"igemm_|igemm1x1_|conv3x3_c64")
IS_WGRAD = re.compile(r"wgrad_bf16_dma_kernel|wgrad_f32|stem_wgrad|wgrad_bf16_kernel")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("creid_pp::", "").replace("igemm_bf16_", "").replace("_kernel", "").replace(", Bf16T", "")[:34]


def blocks(B, H, W):
    """[(block index, {role: shape})] in forward order; shape = (cin, cout, k, stride, hin, win)."""
    sh, out, i, bi = conv_shapes(B, H, W), [], 0, 0
    for _planes, n in zip((64, 128, 256, 512), (3, 4, 6, 3)):
        for b in range(n):
            d = {"c1": sh[i], "c2": sh[i + 1], "c3": sh[i + 2]}
            i += 3
            if b == 0:
                d["ds"] = sh[i]; i += 1
            out.append((bi, d)); bi += 1
    return out


def work(B, shape, what, role=None, bi=None):
    """(FLOPs, algorithmic bytes) of one launch: the model bench.py prices the step with (bench_train.conv_launch_work)."""
    return conv_launch_work(B, shape, what, role, first_block=not bi)


def main(path, B=64, H=256, W=128):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    segs = [rows[a:b] for a, b in zip(marks, marks[1:] + [len(rows)])]
    train = [s for s in segs if any("adam" in r[0] for r in s)]
    if len(train) < 3:
        print("no complete training step in the trace"); return
    step = train[-2]
    convs = [(n, (e - s) / 1e3) for n, s, e in step if IS_CONV.search(n)]
    wgr = [(n, (e - s) / 1e3) for n, s, e in step if IS_WGRAD.search(n)]
    blk = blocks(B, H, W)
    fwd_order = [("stem", None, (3, 64, 7, 2, H, W))]
    for bi, d in blk:
        fwd_order += [(r, bi, d[r]) for r in ("c1", "c2") + (("ds",) if "ds" in d else ()) + ("c3",)]
    bwd_order = []
    for bi, d in reversed(blk):
        bwd_order += [(r, bi, d[r]) for r in ("c3", "c2") + (("ds",) if "ds" in d else ()) + ("c1",)]
    nf = len(fwd_order)
    if len(convs) != nf + len(bwd_order) or len(wgr) not in (len(bwd_order), len(bwd_order) + 1):
        print(f"unexpected launch counts: {len(convs)} convolution launches (expected {nf + len(bwd_order)}), {len(wgr)} weight gradients "
              f"(expected {len(bwd_order)} + stem)")
        return
    table = {}            # (block, role) -> {"shape":..., "fwd": (kernel, us), "dgrad":..., "wgrad":...}
    for (role, bi, shp), (n, t) in zip(fwd_order, convs[:nf]):
        table[(bi, role)] = {"shape": shp, "fwd": (short(n), t)}
    for (role, bi, shp), (n, t) in zip(bwd_order, convs[nf:]):
        table[(bi, role)]["dgrad"] = (short(n), t)
    for (role, bi, shp), (n, t) in zip(bwd_order, wgr):
        table[(bi, role)]["wgrad"] = (short(n), t)
    if len(wgr) == len(bwd_order) + 1:
        table[(None, "stem")]["wgrad"] = (short(wgr[-1][0]), wgr[-1][1])
    print(f"one training step (B = {B}, {H} x {W}): {len(step)} kernels, {sum(e - s for _, s, e in step) / 1e3:.0f} us summed over a "
          f"{(step[-1][2] - step[0][1]) / 1e3:.0f} us span; {len(convs)} forward / data-gradient launches, {len(wgr)} weight gradients\n")
    print("| block | role | shape | pass | kernel | us | TF/s | GB/s (algorithmic) | vs same-shape fwd | speed of light us | x SOL |\n|---|---|---|---|---|---:|---:|---:|---:|---:|---:|")
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    ratios = []
    soltot = {}
    for (bi, role), e in table.items():
        cin, cout, k, st, h, w = e["shape"]
        label = f"{cin}->{cout} k{k} s{st} {h}x{w}"
        for what in ("fwd", "dgrad", "wgrad"):
            if what not in e:
                continue
            kern, t = e[what]
            fl, by = work(B, e["shape"], what, role, bi)
            if role == "stem":
                fl = 2.0 * B * (H // 2) * (W // 2) * 64 * 147
                by = B * (H + 8) * (W + 6) * 4 * 2 + B * (H // 2) * (W // 2) * 64 * 2
            tot[what][0] += fl; tot[what][1] += t
            rel = t / e["fwd"][1]
            if what != "fwd":
                ratios.append((rel, bi, role, label, what, t, fl / t / 1e6))
            sol = max(fl / 2.5e15, by / 6.3e12) * 1e6
            soltot[what] = soltot.get(what, 0.0) + sol
            print(f"| {'' if bi is None else bi} | {role} | {label} | {what} | {kern} | {t:.1f} | {fl / t / 1e6:.0f} | {by / t / 1e3:.0f} | "
                  f"{'' if what == 'fwd' else f'{rel:.2f} x'} | {sol:.1f} | {t / sol:.2f} |")
    print()
    for what, (fl, t) in tot.items():
        print(f"{what}: {t:.0f} us, {fl / t / 1e6:.0f} TF/s over the layer mix = {fl / t / 1e6 / 2500:.3f} of the bf16 MFMA peak")
    print("\nspeed of light per launch = max(2MNK / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s), summed: "
          + ", ".join(f"{k} {v:.0f} us (measured {tot[k][1]:.0f}, x {tot[k][1] / v:.2f})" for k, v in soltot.items())
          + f"; all passes {sum(soltot.values()):.0f} us vs {sum(v[1] for v in tot.values()):.0f} us measured = "
          f"{sum(soltot.values()) / sum(v[1] for v in tot.values()):.3f} of the attainable bound (bench.py roofline.frac_of_sol)")
    print("\nslowest relative to the forward of the same shape:")
    for rel, bi, role, label, what, t, tf in sorted(ratios, reverse=True)[:8]:
        print(f"  block {bi} {role} {label} {what}: {t:.1f} us ({tf:.0f} TF/s) = {rel:.2f} x its forward")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], *(int(v) for v in a[2:5]))
