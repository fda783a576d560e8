"""Build libcreid_hip.so (gfx950) in-tree with hipcc: one object per csrc/*.hip, linked into
centroids-reid_amd/lib/libcreid_hip.so.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcreid_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"]
RES_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, obj: str, verbose: bool, extra=()):
    cmd = [hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    # per-kernel register / scratch / occupancy table next to the object: tests/test_build_cpu.py keeps the hot
    # kernels inside their occupancy budget (a 512-thread igemm workgroup needs <= 128 VGPRs to run two per CU)
    rows, cur, other = [], None, []
    for line in r.stderr.splitlines():
        if "remark:" not in line:
            other.append(line)
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in RES_KEYS:
                cur[RES_KEYS[k.strip()]] = v.strip()
    with open(obj[:-2] + ".res", "w") as f:
        for row in rows:
            f.write(" ".join(f"{k}={v}" for k, v in row.items()) + "\n")
    if other and verbose:
        print("\n".join(other), file=sys.stderr)


def build(force: bool = False, verbose: bool = False, ablation: bool = False) -> str:
    """ablation=True: the timing-ablation build (CREID_IGEMM_ABL / CREID_WGRAD_ABL switches compiled into the k-loops) as
    lib/libcreid_hip_abl.so, for tools/debug/*_abl.py via CREID_LIB_PATH; never loaded by default."""
    if ablation:
        return _build(os.path.join(LIBDIR, "obj_abl"), os.path.join(LIBDIR, "libcreid_hip_abl.so"), force, verbose, ("-DCREID_ABL_BUILD=1",))
    return _build(OBJDIR, LIB, force, verbose, ())


def _build(OBJDIR: str, LIB: str, force: bool, verbose: bool, extra) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s, *hdrs]):
            jobs.append((s, o))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for f in [ex.submit(_compile, s, o, verbose, extra) for s, o in jobs]:
                f.result()
    if jobs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))
