"""Ordered kernel list of ONE replay of the captured training step from a rocprofv3 kernel-trace .db:
index, start offset, duration, gap to the previous kernel's end, grid (workgroups), LDS bytes, name."""
import re
import sqlite3
import sys


def main(path, pat=None):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "image_pad" in r[0]]
    a, b = marks[-3], marks[-2]
    t0 = rows[a][1]; prev_end = t0
    gaps = 0
    for i, (n, s, e, gx, gy, gz, wx, lds) in enumerate(rows[a:b]):
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]
        gap = s - prev_end; gaps += max(gap, 0); prev_end = max(prev_end, e)
        if pat is None or re.search(pat, n):
            print(f"{i:4d} t={(s-t0)/1e3:8.1f} dur={(e-s)/1e3:7.1f} gap={gap/1e3:5.1f} wgs={gx*gy*gz//max(wx,1):6d} lds={lds:6d} {n}")
    print(f"sum of gaps {gaps/1e3:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
