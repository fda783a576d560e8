"""Summarise a rocprofv3 results .db (kernel trace) into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def main(path, out=None, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    tmin, tmax = min(r[1] for r in rows), max(r[2] for r in rows)
    for n, s, e in rows:
        n = re.sub(r"\(.*", "", n)
        n = re.sub(r"^void ", "", n)
        a = agg.setdefault(n, [0, 0])
        a[0] += 1; a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    lines = [f"# rocprofv3 kernel-trace summary: {path}", "",
             f"kernels: {len(rows)} dispatches, busy {tot/1e6:.2f} ms over a {(tmax-tmin)/1e6:.2f} ms window "
             f"({100*tot/(tmax-tmin):.1f}% GPU-busy)", "",
             "| kernel | calls | total ms | avg us | % of busy |", "|---|---:|---:|---:|---:|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        lines.append(f"| {n[:90]} | {c} | {t/1e6:.3f} | {t/c/1e3:.1f} | {100*t/tot:.1f} |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
