"""Per-step kernel table from a rocprofv3 kernel-trace .db of bench.py: isolates ONE replay of the captured training step
(the kernels between the last two image_pad launches) and prints name / launches / total us, plus group totals."""
import re
import sqlite3
import sys

GROUPS = [("conv fwd + dgrad (igemm)", r"igemm_|igemm1x1_|conv3x3_c64|stem_pool|c3_c1_kernel"), ("conv wgrad", r"wgrad_bf16|wgrad_f32"), ("wgrad split reduce", r"wgrad_reduce"),
          ("BN backward apply", r"bn2d_bwd_apply|ibn_bwd_apply"), ("BN apply (incl. finalize + apply in one launch)", r"bn2d_apply|ibn_apply|bn2d_fin_apply"),
          ("BN finalize (fwd+bwd)", r"finalize"), ("BN reduce (unfused)", r"bwd_reduce|col_stats"),
          ("optimisers", r"adam|sgd_scaled|amp_"), ("pool / gap", r"maxpool|gap_"), ("layout", r"weight_prep|image_pad|nhwc"),
          ("heads", r"heads_stage|triplet|center_|xent|bn1d|loo_|gemm_f32|mean_rows|ctl_step")]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    if len(marks) < 3:
        print("not enough steps in the trace"); return
    a, b = marks[-3], marks[-2]                      # a full replay in the middle of the timed loop
    step = rows[a:b]
    span = step[-1][2] - step[0][1]
    agg = {}
    for n, s, e in step:
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:70]
        v = agg.setdefault(n, [0, 0]); v[0] += 1; v[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"one step: {len(step)} kernels, {tot/1e3:.0f} us summed over a {span/1e3:.0f} us span")
    grp = {}
    for n, (c, t) in agg.items():
        g = next((gn for gn, pat in GROUPS if re.search(pat, n)), "torch glue / other")
        v = grp.setdefault(g, [0, 0]); v[0] += c; v[1] += t
    print("| group | launches | us | % |\n|---|---:|---:|---:|")
    for g, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
        print(f"| {g} | {c} | {t/1e3:.0f} | {100*t/tot:.1f} |")
    # BatchNorm apply launches by role (ResNet50: per block c1, c2, [ds on the first block of a layer], c3 in forward order;
    # c3, c2, c1, [ds] per block in reverse block order in backward): what consumer-side fusion of bn1 / bn2 could remove at most
    has_ds = {0, 3, 7, 13}
    fwd_roles = [r for b in range(16) for r in (["c1", "c2"] + (["ds"] if b in has_ds else []) + ["c3"])]
    bwd_roles = [r for b in reversed(range(16)) for r in (["c3", "c2", "c1"] + (["ds"] if b in has_ds else []))]
    for label, pat, roles in (("BN apply (forward)", r"bn2d_apply_kernel", fwd_roles), ("BN backward apply", r"bn2d_bwd_apply_kernel", bwd_roles)):
        ls = [(e - st) / 1e3 for n, st, e in step if re.search(pat, n)]
        if len(ls) not in (len(roles), len(roles) + 1):
            continue
        if len(ls) == len(roles) + 1:                 # backward: the stem's BatchNorm comes last
            roles = roles + ["stem"]
        tot = {}
        for r, t in zip(roles, ls):
            v = tot.setdefault(r, [0, 0.0]); v[0] += 1; v[1] += t
        print(f"\n{label} by role: " + ", ".join(f"{r}: {c} launches {t:.0f} us" for r, (c, t) in sorted(tot.items())))
    print("\n| kernel | launches | us | avg us |\n|---|---:|---:|---:|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"| {n} | {c} | {t/1e3:.0f} | {t/c/1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
