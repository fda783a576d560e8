"""Turn the gpurun_out/pmc_*.json passes (tools/pmc_run.sh) into profiles/<round>_pmc_traffic.json + profiles/<round>_pmc_summary.md (CREID_ROUND, default r05)."""
import json
import os

RND = os.environ.get("CREID_ROUND", "r05")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
meta = json.load(open(os.path.join(G, "pmc_meta.json")))
F, W, S, D = (json.load(open(os.path.join(G, f"pmc_{n}.json"))) for n in ("fetch", "write", "sq", "derived"))
NSIMD = 256 * 4


# every kernel that runs a forward / data-gradient convolution (round 4 added the last three; the family figures of the first
# round-4 report missed them)
IGEMM = ("igemm_bf16_", "creid_pp::igemm_bf16_pp", "igemm1x1_stream", "conv3x3_c64")


def fam(db, prefix, counter):
    prefixes = IGEMM if prefix == "igemm_bf16_" else (prefix,)
    tot = n = 0
    for k, v in db.items():
        if k.startswith(prefixes) and counter in v:
            tot += v[counter]["sum"]; n += v[counter]["dispatches"]
    return tot, n


def first_dispatch(db, prefix):
    """The kernel's first dispatch; the template argument list grew an element type in round 4, so match by prefix."""
    fd = db["_first_dispatch"]
    for k in fd:
        if k.startswith(prefix):
            return fd[k]
    raise KeyError(prefix)


calib = {}
for name, kern in (("calib_l2norm_rows", "l2norm_rows_kernel<0>"), ("calib_bn2d_apply", "bn2d_apply_kernel<unsigned short>"),
                   ("calib_igemm_1x1_stream", "igemm_bf16_ws_kernel<64, 2, 128")):
    f = first_dispatch(F, kern)["FETCH_SIZE"] * 1024
    w = first_dispatch(W, kern)["WRITE_SIZE"] * 1024
    calib[name] = {"known_read_bytes": meta[name]["read"], "known_write_bytes": meta[name]["write"], "FETCH_SIZE_bytes": f,
                   "WRITE_SIZE_bytes": w, "fetch_ratio": f / meta[name]["read"], "write_ratio": w / meta[name]["write"]}
cal_f = first_dispatch(F, "igemm_bf16_ws_kernel<64, 2, 128")["FETCH_SIZE"] * 1024
cal_w = first_dispatch(W, "igemm_bf16_ws_kernel<64, 2, 128")["WRITE_SIZE"] * 1024
out = {"_comment": "HBM-side traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, unit KiB -> bytes), this round, one "
                   "MI355X, tools/pmc_run.sh.  CALIBRATION on kernels of exactly known traffic: FETCH_SIZE reads 0.500x the bytes of "
                   "16-byte streaming loads, both plain global loads (bn2d_apply, bf16) and LDS-DMA (1x1 convolution), WRITE_SIZE "
                   "1.000x -- as MI355X_MICROARCH.md says.  Round 1 calibrated on l2norm_rows, which reads every row TWICE "
                   "(norm pass + scale pass): its 0.99 ratio was 2 reads x 0.5, and the 'no correction' conclusion was wrong.  "
                   "`traffic_bytes` below = 2 * FETCH_SIZE + WRITE_SIZE.",
       "calibration": calib}
for key, prefix in (("igemm_family", "igemm_bf16_"), ("wgrad_family", "wgrad_bf16_")):
    f, n = fam(F, prefix, "FETCH_SIZE"); w, _ = fam(W, prefix, "WRITE_SIZE")
    f *= 1024; w *= 1024
    if key == "igemm_family":
        f -= cal_f; w -= cal_w; n -= 1
    out[key] = {"launches": n, "fetch_bytes": 2 * f, "write_bytes": w, "fetch_size_raw_bytes": f,
                "algorithmic_bytes": meta[key]["algorithmic_bytes"],
                "traffic_over_algorithmic": (2 * f + w) / meta[key]["algorithmic_bytes"]}
def by_prefix(db, key):
    """The entry of a kernel whose template argument list is not part of the key (sqdist_count_f32_kernel<0, true> since round 6)."""
    if key in db:
        return db[key]
    return next(v for k, v in db.items() if k.startswith(key + "<"))


for key in ("sqdist_f32_kernel", "sqdist_count_f32_kernel", "rank_rows_lds_kernel", "stream_poslist_kernel", "cmc_ap_ranked_wide_kernel<false>"):
    f = by_prefix(F, key)["FETCH_SIZE"]["sum"] * 1024; w = by_prefix(W, key)["WRITE_SIZE"]["sum"] * 1024
    e = {"launches": by_prefix(F, key)["FETCH_SIZE"]["dispatches"], "fetch_bytes": 2 * f, "write_bytes": w, "fetch_size_raw_bytes": f}
    if key in meta:
        e["algorithmic_bytes"] = meta[key]["algorithmic_bytes"]
        e["traffic_over_algorithmic"] = (2 * f + w) / meta[key]["algorithmic_bytes"]
    out[key] = e
for key, prefix in (("igemm_family", "igemm_bf16_"), ("wgrad_family", "wgrad_bf16_"), ("sqdist_f32_kernel", "sqdist_f32_kernel"),
                    ("sqdist_count_f32_kernel", "sqdist_count_f32_kernel")):
    mf, _ = fam(S, prefix, "SQ_VALU_MFMA_BUSY_CYCLES"); ga, _ = fam(S, prefix, "GRBM_GUI_ACTIVE")
    out[key]["mfma_busy"] = mf / (ga / 8 * NSIMD)          # matrix-pipe busy fraction by counter (all launches of the family)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_pmc_traffic.json"), "w"), indent=1)

lines = [f"# rocprofv3 --pmc passes, round {RND.lstrip('r0') or '0'} (tools/pmc_run.sh over tools/pmc_kernels.py, one MI355X)", "",
         "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of SIMD-cycles whose matrix",
         "pipe is occupied, i.e. the MFMA-roofline fraction measured by the hardware rather than derived from time.  `MfmaUtil` /",
         "`VALUBusy` are rocprofv3's derived metrics (gfx94x formulas), averaged over the dispatches.", "",
         "| kernel | dispatches | MFMA busy % | MfmaUtil % | VALUBusy % | LDS bank-conflict cycles / SQ busy cycles |", "|---|---:|---:|---:|---:|---:|"]
for k in sorted(S):
    if k.startswith("_") or "at::" in k or "rocclr" in k:
        continue
    v = S[k]
    if "GRBM_GUI_ACTIVE" not in v:
        continue
    simd_cycles = v["GRBM_GUI_ACTIVE"]["sum"] / 8 * NSIMD
    busy = 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / simd_cycles
    d = D.get(k, {})
    mu = d.get("MfmaUtil", {}).get("sum", 0) / max(1, d.get("MfmaUtil", {}).get("dispatches", 1))
    vb = d.get("VALUBusy", {}).get("sum", 0) / max(1, d.get("VALUBusy", {}).get("dispatches", 1))
    bc = v["SQ_LDS_BANK_CONFLICT"]["sum"] / max(1.0, v["SQ_BUSY_CYCLES"]["sum"])
    lines.append(f"| {k} | {v['GRBM_GUI_ACTIVE']['dispatches']} | {busy:.1f} | {mu:.1f} | {vb:.1f} | {bc:.2f} |")
for fam_name, prefix in (("igemm family (conv fwd + dgrad, whole layer mix)", "igemm_bf16_"), ("wgrad family", "wgrad_bf16_")):
    mf, _ = fam(S, prefix, "SQ_VALU_MFMA_BUSY_CYCLES"); ga, n = fam(S, prefix, "GRBM_GUI_ACTIVE")
    lines.append(f"| **{fam_name}** | {n} | **{100 * mf / (ga / 8 * NSIMD):.1f}** | | | |")
ES_path, ED_path = os.path.join(G, "pmc_embed_sq.json"), os.path.join(G, "pmc_embed_derived.json")
if os.path.exists(ES_path):
    ES = json.load(open(ES_path)); ED = json.load(open(ED_path)) if os.path.exists(ED_path) else {}
    lines += ["", "## Eval-mode embedding forward, ResNet50 256 x 128, batch 128 (tools/pmc_embed.py: three forwards, every kernel of the forward)", "",
              "| kernel | dispatches | MFMA busy % | MfmaUtil % | VALUBusy % | LDS bank-conflict cycles / SQ busy cycles |", "|---|---:|---:|---:|---:|---:|"]
    for k in sorted(ES):
        if k.startswith("_") or "at::" in k or "rocclr" in k or "GRBM_GUI_ACTIVE" not in ES[k]:
            continue
        v = ES[k]
        busy = 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (v["GRBM_GUI_ACTIVE"]["sum"] / 8 * NSIMD)
        d = ED.get(k, {})
        mu = d.get("MfmaUtil", {}).get("sum", 0) / max(1, d.get("MfmaUtil", {}).get("dispatches", 1))
        vb = d.get("VALUBusy", {}).get("sum", 0) / max(1, d.get("VALUBusy", {}).get("dispatches", 1))
        bc = v["SQ_LDS_BANK_CONFLICT"]["sum"] / max(1.0, v["SQ_BUSY_CYCLES"]["sum"])
        lines.append(f"| {k} | {v['GRBM_GUI_ACTIVE']['dispatches']} | {busy:.1f} | {mu:.1f} | {vb:.1f} | {bc:.2f} |")
lines += ["", f"HBM-side traffic and the FETCH_SIZE calibration: profiles/{RND}_pmc_traffic.json.", ""]
open(os.path.join(ROOT, "profiles", f"{RND}_pmc_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:4])); print(open(os.path.join(ROOT, "profiles", f"{RND}_pmc_summary.md")).read()[-2500:])
