"""Which launches of one captured training step (or embedding forward) run a PARTIAL LAST ROUND of workgroups?  From a rocprofv3
kernel-trace .db (grid, workgroup size, registers and LDS per dispatch are in it): resident workgroups per launch = what the CU can
hold of that kernel (512 registers per SIMD lane, 160 KB LDS, 32 waves) x 256 CUs; rounds = workgroups / resident.  A launch with
1 < rounds < ~4 and a fractional part pays (ceil(rounds) - rounds) / ceil(rounds) of its duration for a tail on a partly empty chip
-- the pattern behind the streamed evaluation's 588-workgroup grid and the elementwise kernels' fixed 2048 (round 5).
    python tools/grid_tail.py <results.db> [train|embed]"""
import math
import re
import sqlite3
import sys


def main(path, kind="train"):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, lds_size, vgpr_count, "
                       "accum_vgpr_count from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    segs = [rows[a:b] for a, b in zip(marks, marks[1:] + [len(rows)])]
    segs = [s for s in segs if any("adam" in r[0] for r in s) == (kind == "train")]
    if len(segs) < 2:
        print("no complete segment of that kind in the trace"); return
    seg = segs[-2]
    agg = {}
    for n, st, en, gx, gy, gz, wx, wy, wz, lds, vg, ag in seg:
        wg_threads = max(1, wx * wy * wz)
        nwg = (gx * gy * gz) // wg_threads
        waves = (wg_threads + 63) // 64
        regs = max(8, (vg + ag + 7) // 8 * 8)
        w_simd = min(8, 512 // regs)
        by_reg = max(1, (w_simd * 4) // waves)
        by_lds = (160 * 1024) // lds if lds > 0 else 64
        per_cu = max(1, min(by_reg, by_lds, 32 // waves if waves <= 32 else 1))
        resident = per_cu * 256
        rounds = nwg / resident
        us = (en - st) / 1e3
        waste = 0.0
        if rounds > 1.0:
            waste = (math.ceil(rounds) - rounds) / math.ceil(rounds) * us
        short = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]
        key = (short, nwg, per_cu)
        a = agg.setdefault(key, [0, 0.0, 0.0, rounds])
        a[0] += 1; a[1] += us; a[2] += waste
    tot = sum(v[1] for v in agg.values())
    print(f"one {kind} segment: {len(seg)} launches, {tot:.0f} us; launches whose last round is partial, by estimated idle share:")
    print("| kernel | workgroups | resident / CU | rounds | launches | us | est. idle us |\n|---|---:|---:|---:|---:|---:|---:|")
    for (short, nwg, per_cu), (c, us, waste, rounds) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:25]:
        if waste < 1.0:
            continue
        print(f"| {short} | {nwg} | {per_cu} | {rounds:.2f} | {c} | {us:.0f} | {waste:.0f} |")
    print(f"sum of the estimates: {sum(v[2] for v in agg.values()):.0f} us of {tot:.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
