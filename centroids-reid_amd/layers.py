"""Functional wrappers over the stage-A layer kernels (NHWC tensors in the compute dtype).
Used by the unit tests and by small callers; the training schedule itself lives in backbone.py."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _dt(t):
    return L._DT[t.dtype]


def conv_desc(B, H, W, cin, cout, k, stride, pad):
    oh = (H + 2 * pad - k) // stride + 1
    ow = (W + 2 * pad - k) // stride + 1
    return L.ConvDesc(B, H, W, cin, oh, ow, cout, k, k, stride, pad), oh, ow


def weight_prep(w_oihw: torch.Tensor, dtype):
    O, I, kh, kw = w_oihw.shape
    krsc = torch.empty((O, kh, kw, I), dtype=dtype, device=w_oihw.device)
    crsk = torch.empty((I, kh, kw, O), dtype=dtype, device=w_oihw.device)
    L.check(L.lib().creid_weight_prep(L.ptr(w_oihw.contiguous()), O, I, kh, kw, L._DT[dtype], L.ptr(krsc), L.ptr(crsk),
                                      L.stream()), "weight_prep")
    return krsc, crsk


def conv2d_fwd(x, w_krsc, stride, pad, with_stats=False):
    """x [B,H,W,Cin] -> y [B,OH,OW,Cout] (+ partial [rows,2,Cout])."""
    L.require_gpu(x, w_krsc)
    B, H, W, cin = x.shape
    cout, k = w_krsc.shape[0], w_krsc.shape[1]
    d, oh, ow = conv_desc(B, H, W, cin, cout, k, stride, pad)
    y = torch.empty((B, oh, ow, cout), dtype=x.dtype, device=x.device)
    part = None
    if with_stats:
        rows = L.lib().creid_conv2d_bn_partial_rows(C.byref(d))
        part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
    L.check(L.lib().creid_conv2d_fwd_nhwc(C.byref(d), L.ptr(x), L.ptr(w_krsc), L.ptr(y), L.ptr(part), _dt(x), L.stream()),
            "conv2d_fwd")
    return (y, part) if with_stats else y


def bn_fold(gamma, beta, running_mean, running_var, eps=1e-5):
    """Eval-mode BatchNorm as scale_shift float [2, C] (one-entry creid_bn2d_fold_multi table)."""
    import numpy as np
    Cc = running_mean.numel()
    out = torch.empty((2, Cc), dtype=torch.float32, device=running_mean.device)
    rec = np.zeros(1, dtype=np.dtype([("g", "<u8"), ("b", "<u8"), ("m", "<u8"), ("v", "<u8"), ("o", "<u8"), ("C", "<i4"),
                                      ("eps", "<f4")]))
    rec[0] = (gamma.data_ptr() if gamma is not None else 0, beta.data_ptr() if beta is not None else 0,
              running_mean.data_ptr(), running_var.data_ptr(), out.data_ptr(), Cc, eps)
    tab = torch.from_numpy(rec.view(np.uint8).copy()).to(out.device)
    L.check(L.lib().creid_bn2d_fold_multi(L.ptr(tab), 1, L.stream()), "bn2d_fold_multi")
    torch.cuda.current_stream().synchronize()      # `tab` must outlive the launch
    return out


def conv2d_fwd_affine(x, w_krsc, stride, pad, scale_shift, residual=None, relu=True):
    """Eval-mode conv -> BatchNorm(scale_shift [2, Cout]) -> (+residual) -> (ReLU) in one launch."""
    L.require_gpu(x, w_krsc, scale_shift, residual)
    B, H, W, cin = x.shape
    cout, k = w_krsc.shape[0], w_krsc.shape[1]
    d, oh, ow = conv_desc(B, H, W, cin, cout, k, stride, pad)
    y = torch.empty((B, oh, ow, cout), dtype=x.dtype, device=x.device)
    L.check(L.lib().creid_conv2d_fwd_affine_nhwc(C.byref(d), L.ptr(x), L.ptr(w_krsc), L.ptr(y), L.ptr(scale_shift),
                                                 L.ptr(residual), 1 if relu else 0, _dt(x), L.stream()), "conv2d_fwd_affine")
    return y


def conv2d_dgrad(dy, w_crsk, in_hw, stride, pad, add_src=None):
    L.require_gpu(dy, w_crsk)
    B, oh, ow, cout = dy.shape
    cin, k = w_crsk.shape[0], w_crsk.shape[1]
    H, W = in_hw
    d, oh2, ow2 = conv_desc(B, H, W, cin, cout, k, stride, pad)
    assert (oh2, ow2) == (oh, ow)
    dx = torch.empty((B, H, W, cin), dtype=dy.dtype, device=dy.device)
    L.check(L.lib().creid_conv2d_dgrad_nhwc(C.byref(d), L.ptr(dy), L.ptr(w_crsk), L.ptr(dx), L.ptr(add_src), _dt(dy),
                                            L.stream()), "conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, k, stride, pad, out=None, accumulate=False):
    L.require_gpu(x, dy)
    B, H, W, cin = x.shape
    cout = dy.shape[3]
    d, _, _ = conv_desc(B, H, W, cin, cout, k, stride, pad)
    dw = torch.zeros((cout, cin, k, k), dtype=torch.float32, device=x.device) if out is None else out
    nbytes = L.lib().creid_conv2d_wgrad_workspace_bytes(C.byref(d), _dt(x))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    L.check(L.lib().creid_conv2d_wgrad_nhwc(C.byref(d), L.ptr(x), L.ptr(dy), L.ptr(dw), 1 if accumulate else 0, L.ptr(ws),
                                            nbytes, _dt(x), L.stream()), "conv2d_wgrad")
    return dw


def stem_prepare(x_nchw, w_oihw, dtype):
    B, _, H, W = x_nchw.shape
    xpad = torch.empty((B, H + 8, W + 6, 4), dtype=dtype, device=x_nchw.device)
    L.check(L.lib().creid_image_to_nhwc4_pad(L.ptr(x_nchw.contiguous()), B, H, W, L._DT[dtype], L.ptr(xpad), L.stream()),
            "image_pad")
    ws = torch.empty((64, 8, 32), dtype=dtype, device=x_nchw.device)
    L.check(L.lib().creid_stem_weight_prep(L.ptr(w_oihw.contiguous()), L._DT[dtype], L.ptr(ws), L.stream()), "stem_weight_prep")
    return xpad, ws


def stem_conv_fwd(xpad, w_stem, B, H, W):
    y = torch.empty((B, H // 2, W // 2, 64), dtype=xpad.dtype, device=xpad.device)
    L.check(L.lib().creid_stem_conv_fwd(B, H, W, L.ptr(xpad), L.ptr(w_stem), L.ptr(y), None, _dt(xpad), L.stream()),
            "stem_conv_fwd")
    return y


def stem_conv_wgrad(xpad, dy, B, H, W):
    dw = torch.zeros((64, 3, 7, 7), dtype=torch.float32, device=xpad.device)
    nbytes = L.lib().creid_stem_conv_wgrad_workspace_bytes(B, H, W, _dt(xpad))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=xpad.device)
    L.check(L.lib().creid_stem_conv_wgrad(B, H, W, L.ptr(xpad), L.ptr(dy), L.ptr(dw), 0, L.ptr(ws), nbytes, _dt(xpad),
                                          L.stream()), "stem_conv_wgrad")
    return dw


def maxpool_fwd(x):
    B, H, W, Cc = x.shape
    y = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.uint8, device=x.device)
    L.check(L.lib().creid_maxpool3x3s2_fwd(L.ptr(x), B, H, W, Cc, _dt(x), L.ptr(y), L.ptr(idx), L.stream()), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, H, W):
    B, _, _, Cc = dy.shape
    dx = torch.empty((B, H, W, Cc), dtype=dy.dtype, device=dy.device)
    L.check(L.lib().creid_maxpool3x3s2_bwd(L.ptr(dy), L.ptr(idx), B, H, W, Cc, _dt(dy), L.ptr(dx), L.stream()), "maxpool_bwd")
    return dx


def bn2d_train_fwd(x, gamma, beta, rmean, rvar, residual=None, relu=True, momentum=0.1, eps=1e-5, partial=None):
    """x [.., C] NHWC -> (y, mean, invstd); statistics from `partial` (conv epilogue) or a column pass."""
    Cc = x.shape[-1]
    M = x.numel() // Cc
    lib, st = L.lib(), L.stream()
    if partial is None:
        rows = lib.creid_col_stats_rows(M)
        partial = torch.empty((rows, 2, Cc), dtype=torch.float32, device=x.device)
        L.check(lib.creid_col_stats(L.ptr(x), M, Cc, _dt(x), L.ptr(partial), st), "col_stats")
    rows = partial.shape[0]
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    ss = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    L.check(lib.creid_bn2d_finalize(L.ptr(partial), rows, Cc, M, L.ptr(rmean), L.ptr(rvar), 1, momentum, eps, L.ptr(gamma),
                                    L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(ss), st), "bn2d_finalize")
    y = torch.empty_like(x)
    L.check(lib.creid_bn2d_apply(L.ptr(x), L.ptr(ss), L.ptr(residual), 1 if relu else 0, M, Cc, _dt(x), L.ptr(y), st),
            "bn2d_apply")
    return y, mean, invstd


def bn2d_bwd(x, g, act, mean, invstd, gamma, want_gm=False):
    Cc = x.shape[-1]
    M = x.numel() // Cc
    lib, st = L.lib(), L.stream()
    rows = lib.creid_bn2d_bwd_rows(M)
    part = torch.empty((rows, 2, Cc), dtype=torch.float32, device=x.device)
    sums = torch.empty((3, Cc), dtype=torch.float32, device=x.device)
    dgamma = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.zeros_like(dgamma)
    dx = torch.empty_like(x)
    gm = torch.empty_like(x) if want_gm else None
    L.check(lib.creid_bn2d_bwd(L.ptr(x), L.ptr(g), L.ptr(act), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), M, Cc, _dt(x),
                               L.ptr(part), 0, L.ptr(sums), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dx), L.ptr(gm), st), "bn2d_bwd")
    return dx, dgamma, dbeta, gm


def conv2d_dgrad_bnred(dy, w_crsk, in_hw, stride, pad, bn_x, bn_act, mean, invstd, add_src=None, add_src_stride=1):
    """dgrad + fused column reduction of the next BN backward -> (dx, partial [rows, 2, Cin])."""
    L.require_gpu(dy, w_crsk, bn_x)
    B, oh, ow, cout = dy.shape
    cin, k = w_crsk.shape[0], w_crsk.shape[1]
    H, W = in_hw
    d, _, _ = conv_desc(B, H, W, cin, cout, k, stride, pad)
    dx = torch.empty((B, H, W, cin), dtype=dy.dtype, device=dy.device)
    rows = L.lib().creid_bn2d_bwd_rows(B * H * W)
    part = torch.empty((rows, 2, cin), dtype=torch.float32, device=dy.device)
    L.check(L.lib().creid_conv2d_dgrad_bnred_nhwc(C.byref(d), L.ptr(dy), L.ptr(w_crsk), L.ptr(dx), L.ptr(add_src), L.ptr(bn_x),
                                                  L.ptr(bn_act), L.ptr(mean), L.ptr(invstd), L.ptr(part), 0, add_src_stride, _dt(dy), L.stream()),
            "conv2d_dgrad_bnred")
    return dx, part
