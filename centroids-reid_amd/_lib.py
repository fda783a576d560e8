"""ctypes binding of libcreid_hip.so (the C ABI declared in include/creid.h).

The product path has NO CPU fallback: if the shared library is missing this module raises,
and every op raises if its tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CREID_LIB_PATH") or os.path.join(_HERE, "lib", "libcreid_hip.so")   # override: A/B of two builds

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

_lib = None

_p, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/creid.h declares
SIGNATURES = {
    "creid_abi_version": (C.c_int, []),
    "creid_l2norm_rows": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _f32, _p]),
    "creid_row_sqnorm": (C.c_int, [_p, _p, _i64, _i64, C.c_int, _p]),
    "creid_sqdist_matrix": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p, _i64, _p]),
    "creid_rank_rows_workspace_bytes": (_sz, [_i64, _i64]),
    "creid_rank_rows": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "creid_rank_rows_eval": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _sz, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_cmc_ap_ranked": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_cmc_ap_ranked_camsets": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_topk_rows": (C.c_int, [_p, _i64, _i64, _i64, _i32, _p, _p, _p, _p]),
    "creid_eval_reduce": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "creid_stream_plan": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "creid_stream_poslist": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "creid_stream_count": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "creid_stream_finalize": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p]),
    "creid_tune_set": (C.c_int, [_i32, _i64, _i64, _i64, _i64, _i32, _i32, _i32]),
    "creid_tune_clear": (C.c_int, []),
    "creid_loo_centroids_fwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "creid_loo_centroids_bwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "creid_loo_emb_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p]),
    "creid_loo_emb_bwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "creid_ctl_step_stats": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p]),
    "creid_loo_emb_fwd_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "creid_loo_emb_fwd_rows_lonely": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_triplet_fwd_batched_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _f32, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "creid_ctl_round_scale": (C.c_int, [_p, _i64, _p, _p]),
    "creid_ctl_heads_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "creid_ctl_heads_fused": (C.c_int, [_p, _p]),
    "creid_ctl_step_stats_rows": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p]),
    "creid_center_loss_fwd_masked": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "creid_center_loss_bwd_masked": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _p, _f32, _p, _p, _p]),
    "creid_xent_ls_masked": (C.c_int, [_p, _p, _p, _i64, _i64, _f32, _f32, _p, _p, _p, _p]),
    "creid_bn1d_fwd_masked": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p]),
    "creid_bn1d_bwd_masked": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "creid_triplet_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _f32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_triplet_bwd": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _f32, _p, _p]),
    "creid_triplet_fwd_batched": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _f32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_triplet_bwd_batched": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _f32, _p, _p]),
    "creid_triplet_cosine_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _f32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_triplet_cosine_bwd": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _f32, _p, _p]),
    "creid_rownorm_fwd": (C.c_int, [_p, _i64, _i64, C.c_int, _f32, _p, _p, _p]),
    "creid_rownorm_bwd": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _f32, _p, _p]),
    "creid_hard_mine_from_dist": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _p]),
    "creid_clamp_sqrt_inplace": (C.c_int, [_p, _i64, _f32, _p]),
    "creid_center_loss_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "creid_center_loss_bwd": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _f32, _p, _p, _p]),
    "creid_xent_ls": (C.c_int, [_p, _p, _i64, _i64, _f32, _f32, _p, _p, _p, _p]),
    "creid_bn1d_fwd": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, C.c_int, _f32, _f32, _p, _p, _p, _p]),
    "creid_bn1d_bwd": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "creid_gather_mean_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "creid_adam_step": (C.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _p]),
    "creid_adam_step_dev": (C.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _f32, _f32, _f32, _f32, _p]),
    "creid_sgd_scaled_step": (C.c_int, [_p, _p, _i64, _f32, _f32, _p]),
    "creid_amp_scale": (C.c_int, [_p, _i64, _p, _p, _p]),
    "creid_amp_unscale_check": (C.c_int, [_p, _i64, _p, _p, _p]),
    "creid_amp_update": (C.c_int, [_p, _p, _f32, _f32, C.c_int32, _p]),
    "creid_adam_step_dev_amp": (C.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _f32, _f32, _f32, _f32, _p, _p]),
    "creid_sgd_scaled_step_amp": (C.c_int, [_p, _p, _i64, _f32, _f32, _p, _p]),
    "creid_conv2d_bn_partial_rows": (_i64, [_p]),
    "creid_conv2d_fwd_nhwc": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _p]),
    "creid_conv2d_fwd_affine_nhwc": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "creid_conv1x1_bnrelu_fwd": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, C.c_int, _p]),
    "creid_bn2d_fold_entry_bytes": (_i64, []),
    "creid_bn2d_fold_multi": (C.c_int, [_p, _i64, _p]),
    "creid_stem_conv_fwd_affine": (C.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "creid_stem_conv_pool_fwd_affine": (C.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, C.c_int, C.c_int, _p]),
    "creid_bottleneck_c3_c1_fwd_affine": (C.c_int, [_i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p]),
    "creid_bottleneck_c3_c1_fwd_stats": (C.c_int, [_i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p]),
    "creid_conv2d_dgrad_nhwc": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _p]),
    "creid_conv2d_wgrad_workspace_bytes": (_sz, [_p, C.c_int]),
    "creid_conv2d_wgrad_nhwc": (C.c_int, [_p, _p, _p, _p, C.c_int, _p, _sz, C.c_int, _p]),
    "creid_conv2d_wgrad_partials": (C.c_int, [_p, _p, _p, _p, _sz, C.c_int, _p]),
    "creid_conv2d_wgrad_partials_bnfin": (C.c_int, [_p, _p, _p, _p, _sz, C.c_int, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "creid_conv2d_wgrad_reduce": (C.c_int, [_p, _p, C.c_int, _p, _sz, C.c_int, _p]),
    "creid_bn2d_bwd_finalize_wred": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p, _sz, C.c_int, _p]),
    "creid_conv2d_wgrad_reduce_job": (C.c_int, [_p, _p, C.c_int, _p, _sz, C.c_int, _p]),
    "creid_stem_conv_fwd": (C.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, C.c_int, _p]),
    "creid_stem_conv_wgrad_workspace_bytes": (_sz, [_i64, _i64, _i64, C.c_int]),
    "creid_stem_conv_wgrad": (C.c_int, [_i64, _i64, _i64, _p, _p, _p, C.c_int, _p, _sz, C.c_int, _p]),
    "creid_image_to_nhwc4_pad": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p]),
    "creid_augment_u8": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64] + [C.c_float] * 9 + [C.c_int32, C.c_int32, _p, _p]),
    "creid_weight_prep": (C.c_int, [_p, _i64, _i64, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_weight_prep_entry_bytes": (_i64, []),
    "creid_weight_prep_multi": (C.c_int, [_p, _p, _i64, _i64, C.c_int, _p]),
    "creid_stem_weight_prep": (C.c_int, [_p, C.c_int, _p, _p]),
    "creid_bn2d_finalize": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, C.c_int, _f32, _f32, _p, _p, _p, _p, _p, _p]),
    "creid_col_stats_rows": (_i64, [_i64]),
    "creid_col_stats": (C.c_int, [_p, _i64, _i64, C.c_int, _p, _p]),
    "creid_bn2d_apply": (C.c_int, [_p, _p, _p, C.c_int, _i64, _i64, C.c_int, _p, _p]),
    "creid_bn2d_apply_mask": (C.c_int, [_p, _p, _p, C.c_int, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_bn2d_finalize_apply_mask": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_int, _p]),
    "creid_bn2d_apply_dual_mask": (C.c_int, [_p, _p, _p, _p, C.c_int, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_bn2d_bwd_rows": (_i64, [_i64]),
    "creid_bn2d_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p]),
    "creid_bn2d_bwd_mask": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p]),
    "creid_bn2d_bwd_mask_reduce2": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_conv2d_dgrad_bnred_nhwc": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, C.c_int, C.c_int, _p]),
    "creid_conv2d_dgrad_fused_nhwc": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, C.c_int, _p, _sz,
                                                C.c_int, _p]),
    "creid_ibn_rows_per_image": (_i64, [_i64]),
    "creid_ibn_fwd": (C.c_int, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, C.c_int, _f32, _f32, C.c_int, C.c_int,
                                _p, C.c_int, _p, _p, _p, _p, _p]),
    "creid_ibn_bwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p,
                                _p, _p, _p]),
    "creid_ibn_fwd_mask": (C.c_int, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, C.c_int, _f32, _f32, C.c_int, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p]),
    "creid_ibn_bwd_mask": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    "creid_bn2d_apply_maxpool3x3s2": (C.c_int, [_p, _p, C.c_int, _i64, _i64, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_bn2d_bwd_pooled": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, C.c_int, _p, _p, _p, _p, _p, _p]),
    "creid_maxpool3x3s2_fwd": (C.c_int, [_p, _i64, _i64, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_maxpool3x3s2_bwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, C.c_int, _p, _p]),
    "creid_gap_fwd": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p]),
    "creid_gap_fwd_count": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p, _p]),
    "creid_gap_bwd": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p]),
    "creid_nhwc_to_nchw_f32": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p]),
    "creid_gemm_f32": (C.c_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _f32, _f32, _i32, _p]),
}


class ConvDesc(C.Structure):
    """creid_conv_desc of include/creid.h."""
    _fields_ = [("batch", _i64), ("in_h", _i64), ("in_w", _i64), ("in_c", _i64), ("out_h", _i64), ("out_w", _i64),
                ("out_c", _i64), ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32)]


class CtlHeads(C.Structure):
    """creid_ctl_heads of include/creid.h (tests/test_abi_cpu.py compares the two field lists)."""
    _fields_ = ([(n, _i64) for n in ("B", "P", "K", "D", "num_classes", "num_centers", "HW")] +
                [(n, _i32) for n in ("g_dtype", "masked", "split_logits", "split_dbnf")] +
                [(n, _f32) for n in ("margin", "xent_eps", "w_query", "w_center", "w_xent", "w_centroid", "bn_momentum", "bn_eps")] +
                [(n, _p) for n in ("feat", "labels", "is_real", "centers", "bn_weight", "bn_bias", "bn_running_mean",
                                   "bn_running_var", "fc_weight", "loss_weights", "amp_state", "d_centers", "d_bn_weight",
                                   "d_bn_bias", "d_fc_weight", "bn_batches_tracked", "lonely", "stats", "g", "dfeat_out",
                                   "workspace")] +
                [("workspace_bytes", _sz)] +
                [(n, _p) for n in ("bn_x", "bn_mask", "bn_mean", "bn_invstd", "bn_partial")])


class CreidError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CreidError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.creid_abi_version() != 1:
            raise CreidError("libcreid_hip.so ABI version mismatch")
        _lib = h
        global N_PLANS
        N_PLANS = load_tuned_plans()
    return _lib


N_PLANS = 0        # launch plans registered from tuned_plans.json (0: built-in tile / split / ring-depth rules only)


PLANS_PATH = os.path.join(_HERE, "tuned_plans.json")


def load_tuned_plans(path=None):
    """Register the measured per-shape launch plans (tools/tune_plans.py) with the library; CREID_TUNED_PLANS=0 skips
    them (built-in rules everywhere).  Returns the number of plans registered."""
    import json
    sel = os.environ.get("CREID_TUNED_PLANS", "1")
    if sel == "0":
        return 0
    path = path or (sel if sel not in ("", "1") else PLANS_PATH)      # a path: A/B of two plan files
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        plans = json.load(f).get("plans", [])
    for e in plans:
        _lib.creid_tune_set(int(e["kind"]), *[int(v) for v in e["key"]], *[int(v) for v in e["plan"]])
    return len(plans)


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise CreidError(f"unsupported dtype {t.dtype}") from None


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CreidError("centroids-reid_amd ops need HIP device tensors (no CPU fallback); got a CPU tensor")
        if t is not None and not t.is_contiguous():
            raise CreidError("centroids-reid_amd ops need contiguous tensors")


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported dtype", -3: "workspace too small", -4: "unsupported shape"}.get(
            rc, f"hipError_t {rc}")
        raise CreidError(f"{what} failed: {kind}")
