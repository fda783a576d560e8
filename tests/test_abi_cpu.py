"""CPU: the C-ABI library builds, loads, and exports every symbol include/creid.h declares.
No compute calls here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as ge
    ge.build()
    import centroids_reid_amd._lib as L
    assert os.path.exists(L.LIB_PATH)
    return L


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "creid.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(creid_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(built_lib):
    h = ctypes.CDLL(built_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/creid.h but not exported"


def test_binding_covers_header(built_lib):
    assert sorted(built_lib.SIGNATURES) == declared_symbols()
    assert built_lib.lib().creid_abi_version() == 1


def test_binding_arity_and_pointer_slots_match_header(built_lib):
    """Every ctypes signature has the header's parameter count, pointers where the header has pointers and
    64-bit integers where it has int64_t / size_t (a stale binding would pass garbage silently)."""
    import ctypes as C
    txt = open(os.path.join(ROOT, "include", "creid.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = re.findall(r"\b(?:int|int64_t|size_t)\s+(creid_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
    assert len(decls) == len(declared_symbols())
    for name, params in decls:
        plist = [q.strip() for q in params.split(",")]
        if plist == ["void"] or plist == [""]:
            plist = []
        restype, argtypes = built_lib.SIGNATURES[name]
        assert len(argtypes) == len(plist), (name, len(argtypes), len(plist))
        for a, q in zip(argtypes, plist):
            is_ptr = "*" in q
            assert (a is C.c_void_p) == is_ptr, (name, q, a)
            if not is_ptr and re.search(r"\b(int64_t|size_t)\b", q):
                assert C.sizeof(a) == 8, (name, q, a)
            if not is_ptr and re.search(r"\bfloat\b", q):
                assert a is C.c_float, (name, q, a)


def test_ctl_heads_struct_matches_header(built_lib):
    """creid_ctl_heads is passed by pointer: the ctypes Structure must list the header's fields in the header's order with the
    header's widths (a stale copy would shift every pointer behind the first difference)."""
    import ctypes as C
    txt = open(os.path.join(ROOT, "include", "creid.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef struct creid_ctl_heads \{(.*?)\} creid_ctl_heads;", txt, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ptr = "*" in decl
        base = re.match(r"(?:const\s+)?(\w+)", decl).group(1)
        names = [re.sub(r"[\s*]", "", q) for q in decl[decl.index(base) + len(base):].split(",")]
        for nme in names:
            fields.append((nme, ptr, base))
    got = built_lib.CtlHeads._fields_
    assert [f[0] for f in got] == [f[0] for f in fields]
    width = {"int64_t": 8, "int32_t": 4, "float": 4, "size_t": C.sizeof(C.c_size_t)}
    for (name, ctype), (_, ptr, base) in zip(got, fields):
        if ptr:
            assert ctype is C.c_void_p, name
        else:
            assert C.sizeof(ctype) == width[base], name
            assert (ctype is C.c_float) == (base == "float"), name


def test_header_is_plain_c():
    """include/creid.h is the FFI boundary: it must compile as C99 on its own (no torch / HIP / C++ types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "creid.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback(built_lib):
    import torch
    from centroids_reid_amd import reid_metric as rm
    with pytest.raises(built_lib.CreidError):
        rm.get_euclidean(torch.zeros(4, 8), torch.zeros(4, 8))
    with pytest.raises(built_lib.CreidError):
        rm.R1_mAP(num_query=1).compute(torch.zeros(4, 8), [0, 0, 1, 1], [0, 1, 0, 1])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "centroids-reid_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_tuned_plan_file_is_well_formed_and_registers():
    """centroids-reid_amd/tuned_plans.json: every entry has a known kind, a 4-slot key, a 3-slot plan with tile sizes the
    kernels instantiate, and registers through creid_tune_set (host-side registry, no GPU needed)."""
    import json
    from centroids_reid_amd import _lib as L
    path = os.path.join(os.path.dirname(L.__file__), "tuned_plans.json")
    plans = json.load(open(path))["plans"]
    assert len(plans) >= 20
    seen = set()
    for e in plans:
        kind, key, plan = e["kind"], e["key"], e["plan"]
        assert kind in (0, 1) and len(key) == 4 and len(plan) == 3 and all(isinstance(v, int) for v in key + plan)
        assert (kind, tuple(key)) not in seen, "duplicate plan key"
        seen.add((kind, tuple(key)))
        if kind == 0:                                   # weight gradient: (tile rows, tile cols, splits | depth << 16 | ws << 20)
            assert plan[0] in (64, 128) and plan[1] in (64, 128) and 1 <= (plan[2] & 0xffff) <= 4096
            assert ((plan[2] >> 16) & 0xf) in (0, 2, 3, 4) and key[3] in (2, 4)   # stride << 1
        elif plan[2] == 5:                              # all-waves-multiply persistent kernel (conv_pipe.hip): plan[0] = variant word
            v = plan[0]
            bm, bn, kph, mode = (v & 3) * 128, ((v >> 2) & 3) * 128, (v >> 4) & 7, (v >> 8) & 3
            assert (bm, bn) in ((256, 256), (128, 256), (256, 128), (128, 128)) and kph in (1, 2) and mode in (0, 2) and plan[1] == 0
            assert key[1] % bn == 0 and key[2] % 64 == 0 and (key[3] & ~8) in (2, 4)      # forward only; bit 3 = eval-mode epilogue
        else:                                           # forward / data gradient: (N tile, ring depth, kernel kind)
            assert plan[0] in (64, 128) and plan[1] in (2, 3, 4) and plan[2] in (0, 1, 2, 3, 4)
            if kind == 1 and plan[2] == 3:                  # 256-row tiles exist as <128, 2 | 3, 256> only
                assert plan[0] == 128 and plan[1] in (2, 3)
            if kind == 1 and plan[2] == 4:                  # second persistent 1x1 kernel: K = 64 | 128 | 256, forward only
                assert plan[0] == 128 and key[2] in (64, 128, 256) and key[3] == 2
            assert key[1] % plan[0] == 0 and key[3] in (2, 3, 4, 5)          # transposed | stride << 1
    lib = L.lib()
    assert L.load_tuned_plans(path) == len(plans)
    lib.creid_tune_clear()
    assert L.load_tuned_plans() == len(plans)           # back to the shipped set
