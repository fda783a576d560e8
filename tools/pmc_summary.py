"""Aggregate a rocprofv3 --pmc results .db: per (short kernel name, counter) the summed value and the dispatch count.
    python tools/pmc_summary.py <results.db> [out.json]"""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:80]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, value, dispatch_id, start, end from counters_collection").fetchall()
    agg, order = {}, {}
    for kn, cn, v, did, st, en in rows:
        k = short(kn)
        a = agg.setdefault(k, {})
        c = a.setdefault(cn, [0.0, set(), 0])
        c[0] += float(v); c[1].add(did)
        order.setdefault(k, did)
    res = {k: {cn: {"sum": c[0], "dispatches": len(c[1])} for cn, c in a.items()} for k, a in agg.items()}
    # per-dispatch list for the calibration kernels (first dispatch of each kernel name)
    first = {}
    for kn, cn, v, did, st, en in rows:
        k = short(kn)
        if did == order[k]:
            first.setdefault(k, {})[cn] = float(v)
            first[k]["duration_ns"] = en - st
    res["_first_dispatch"] = first
    if out:
        json.dump(res, open(out, "w"), indent=1)
    for k in sorted(res):
        if k.startswith("_") or "at::" in k or "rocclr" in k:
            continue
        print(k, {cn: (round(v["sum"], 1), v["dispatches"]) for cn, v in res[k].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
