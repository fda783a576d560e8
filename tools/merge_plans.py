"""Additive merge of launch-plan files: keep every entry of BASE, append the entries of NEW whose (kind, key) BASE does not have.
tools/tune_plans.py --merge REPLACES the keys it re-measures -- right for re-tuning one batch, wrong when a new batch shares
(M, N, K, stride) keys with a validated one (a training forward and an eval forward of different batches can): plans validated in
situ must not change under a later tuner run for another workload.
    python tools/merge_plans.py BASE.json NEW.json OUT.json"""
import json
import sys

base, new = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
have = {(e["kind"], tuple(e["key"])) for e in base["plans"]}
add = [e for e in new["plans"] if (e["kind"], tuple(e["key"])) not in have]
out = dict(base)
out["plans"] = base["plans"] + add
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(base['plans'])} kept + {len(add)} added = {len(out['plans'])} plans -> {sys.argv[3]}")
