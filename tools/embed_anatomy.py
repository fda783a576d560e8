"""Per-launch table of ONE replay of the eval-mode embedding forward from a rocprofv3 kernel-trace .db of `bench.py --inner-trace`
(segments between image_pad launches that contain no Adam kernel): every convolution labelled with its layer shape, its
duration, achieved TFLOP/s and the HBM rate of its algorithmic bytes (input + output (+ residual) activations, bf16).
    python tools/embed_anatomy.py <results.db> [batch H W]"""
import re
import sqlite3
import sys

sys.path.insert(0, ".")
from centroids_reid_amd.bench_train import conv_shapes   # noqa: E402


def main(path, B=128, H=256, W=128):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    segs = [rows[a:b] for a, b in zip(marks, marks[1:] + [len(rows)])]
    emb = [s for s in segs if not any("adam" in r[0] for r in s)]
    if len(emb) < 2:
        print("no embedding segment in the trace"); return
    seg = emb[-2]
    shapes = conv_shapes(B, H, W)
    # forward launch order per block: c1, c2, (ds), c3 ; conv_shapes order: c1, c2, c3, (ds)
    order, i = [], 0
    while i < len(shapes):
        blk = shapes[i:i + 3]; i += 3
        ds = None
        if i < len(shapes) and shapes[i][2] == 1 and shapes[i][1] == blk[2][1] and shapes[i][0] == blk[0][0] and (len(order) == 0 or shapes[i][0] != shapes[i][1]):
            # downsample entry follows the first block of a layer
            ds = shapes[i]; i += 1
        order += [("c1", blk[0]), ("c2", blk[1])] + ([("ds", ds)] if ds else []) + [("c3", blk[2])]
    is_conv = lambda n: "igemm" in n or "conv3x3_c64" in n or "stem_pool" in n or "c3_c1_kernel" in n   # kernels that run convolutions
    convs = [r for r in seg if is_conv(r[0])]
    n_pair = sum(1 for r in convs if "c3_c1_kernel" in r[0])
    print(f"one embedding forward: {len(seg)} kernels, {sum(e - s for _, s, e in seg) / 1e3:.0f} us summed, span {(seg[-1][2] - seg[0][1]) / 1e3:.0f} us; "
          f"{len(convs)} convolution launches running {len(convs) + n_pair} convolutions (expected {len(order) + 1})")
    print("| # | role | shape | kernel | us | TF/s | GB/s (algorithmic) |\n|---|---|---|---|---:|---:|---:|")
    tot_fl = tot_t = 0.0
    ci = 0
    for n, s, e in seg:
        t = (e - s) / 1e3
        short = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:46]
        if is_conv(n):
            if ci == 0:
                fl = 2.0 * B * (H // 2) * (W // 2) * 64 * 147
                by = B * (H + 8) * (W + 6) * 4 * 2 + B * (H // 2) * (W // 2) * 64 * 2
                if "stem_pool" in n:                  # one launch with the max-pool: only the pooled tensor is written
                    by = B * (H + 8) * (W + 6) * 4 * 2 + B * (H // 4) * (W // 4) * 64 * 2
                role, label = "stem" + (" + pool" if "stem_pool" in n else ""), f"3->64 k7 s2 {H}x{W}"
            elif "c3_c1_kernel" in n:
                # conv3 of this block and conv1 of the next in one launch (conv_pair.hip): two entries of the layer order
                (_, (cin, cout, k, st, h, w)), (_, (cin2, cout2, _, _, _, _)) = order[ci - 1], order[ci]
                fl = 2.0 * B * h * w * (cout * cin + cout2 * cin2)
                by = B * h * w * (cin + 2 * cout + cout2) * 2
                role, label = "c3 + next c1", f"{cin}->{cout}->{cout2} k1 s1 {h}x{w}"
                ci += 1
            else:
                role, (cin, cout, k, st, h, w) = order[ci - 1]
                ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
                fl = 2.0 * B * ho * wo * cout * cin * k * k
                by = (B * h * w * cin + B * ho * wo * cout * (2 if role == "c3" else 1)) * 2
                label = f"{cin}->{cout} k{k} s{st} {h}x{w}"
            ci += 1
            tot_fl += fl; tot_t += t
            print(f"| {ci} | {role} | {label} | {short} | {t:.1f} | {fl / t / 1e6:.0f} | {by / t / 1e3:.0f} |")
        else:
            print(f"|  | - | | {short} | {t:.1f} | | |")
    print(f"\nconvolution launches: {tot_t:.0f} us, {tot_fl / tot_t / 1e6:.0f} TF/s over the forward")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], *(int(v) for v in a[2:5]))
