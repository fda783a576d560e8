"""torch.autograd glue over the C ABI for the head kernels (stages B/C + BNNeck + classifier).

Each Function calls libcreid_hip through ctypes on the current torch stream; PyTorch only
provides device memory, the stream and the autograd tape (no torch compute ops on the path,
except index/cat glue in the callers)."""
from __future__ import annotations

import torch

import os

from . import _lib as L

_DETERMINISTIC = os.environ.get("CREID_DETERMINISTIC", "0") == "1"


def _u8(mask):
    if mask is None:
        return None
    return mask.to(torch.uint8).contiguous()


def _f32c(t):
    t = t.contiguous()
    if t.dtype != torch.float32:
        t = t.float()
    return t


# --------------------------------------------------------------------------- centroids
class LooCentroids(torch.autograd.Function):
    """train_ctl_model.py:79-104 -> (centroids [K,P,D], valid_inst int32 [K,P])."""

    @staticmethod
    def forward(ctx, feat, is_real, P, K):
        feat = _f32c(feat)
        L.require_gpu(feat, is_real)
        ir = _u8(is_real)
        D = feat.shape[1]
        assert feat.shape[0] == P * K
        cent = torch.empty((K, P, D), dtype=torch.float32, device=feat.device)
        valid = torch.empty((K, P), dtype=torch.int32, device=feat.device)
        L.check(L.lib().creid_loo_centroids_fwd(L.ptr(feat), L.ptr(ir), P, K, D, L.ptr(cent), L.ptr(valid),
                                                L.stream()), "creid_loo_centroids_fwd")
        ctx.save_for_backward(ir)
        ctx.dims = (P, K, D)
        ctx.mark_non_differentiable(valid)
        return cent, valid

    @staticmethod
    def backward(ctx, dcent, _dvalid):
        (ir,) = ctx.saved_tensors
        P, K, D = ctx.dims
        dfeat = torch.zeros((P * K, D), dtype=torch.float32, device=dcent.device)
        L.check(L.lib().creid_loo_centroids_bwd(L.ptr(_f32c(dcent)), L.ptr(ir), P, K, D, L.ptr(dfeat), L.stream()),
                "creid_loo_centroids_bwd")
        return dfeat, None, None, None


# --------------------------------------------------------------------------- triplet
class TripletHardMine(torch.autograd.Function):
    """losses/triplet_loss.py:139-173 -> (loss, dist_ap[N], dist_an[N], stats4)."""

    @staticmethod
    def forward(ctx, x, labels, mask, margin):
        x = _f32c(x)
        labels = labels.to(torch.int64).contiguous()
        L.require_gpu(x, labels)
        N, D = x.shape
        dev = x.device
        m8 = _u8(mask)
        dap = torch.empty(N, dtype=torch.float32, device=dev)
        dan = torch.empty(N, dtype=torch.float32, device=dev)
        pi = torch.empty(N, dtype=torch.int32, device=dev)
        ni = torch.empty(N, dtype=torch.int32, device=dev)
        coef = torch.empty(N, dtype=torch.float32, device=dev)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        L.check(L.lib().creid_triplet_fwd(L.ptr(x), L.ptr(labels), L.ptr(m8), N, D,
                                          float(margin) if margin is not None else -1.0, L.ptr(dap), L.ptr(dan),
                                          L.ptr(pi), L.ptr(ni), L.ptr(coef), L.ptr(out4), None, L.stream()),
                "creid_triplet_fwd")
        ctx.save_for_backward(x, dap, dan, pi, ni, coef)
        ctx.mark_non_differentiable(dap, dan, out4)
        return out4[0].clone(), dap, dan, out4

    @staticmethod
    def backward(ctx, gloss, _gap, _gan, _g4):
        x, dap, dan, pi, ni, coef = ctx.saved_tensors
        N, D = x.shape
        dx = torch.zeros_like(x)
        g = _f32c(gloss.reshape(1))
        L.check(L.lib().creid_triplet_bwd(L.ptr(x), N, D, L.ptr(dap), L.ptr(dan), L.ptr(pi), L.ptr(ni), L.ptr(coef),
                                          L.ptr(g), 1.0, L.ptr(dx), L.stream()), "creid_triplet_bwd")
        return dx, None, None, None


def pairwise_dist_mine(x, labels):
    """(dist_mat [N,N], dist_ap, dist_an, p_inds, n_inds): losses/triplet_loss.py:27-41,68-119."""
    x = _f32c(x)
    labels = labels.to(torch.int64).contiguous()
    L.require_gpu(x, labels)
    N, D = x.shape
    dev = x.device
    dap = torch.empty(N, dtype=torch.float32, device=dev); dan = torch.empty_like(dap)
    pi = torch.empty(N, dtype=torch.int32, device=dev); ni = torch.empty_like(pi)
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    dm = torch.empty((N, N), dtype=torch.float32, device=dev)
    L.check(L.lib().creid_triplet_fwd(L.ptr(x), L.ptr(labels), None, N, D, 0.0, L.ptr(dap), L.ptr(dan), L.ptr(pi),
                                      L.ptr(ni), None, L.ptr(out4), L.ptr(dm), L.stream()), "creid_triplet_fwd")
    return dm, dap, dan, pi.long(), ni.long()


# --------------------------------------------------------------------------- center loss
class CenterLossFn(torch.autograd.Function):
    """losses/center_loss.py:26-46."""

    @staticmethod
    def forward(ctx, x, labels, centers):
        x = _f32c(x); centers = _f32c(centers)
        labels = labels.to(torch.int64).contiguous()
        L.require_gpu(x, labels, centers)
        B, D = x.shape
        C = centers.shape[0]
        row = torch.empty(B, dtype=torch.float32, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        L.check(L.lib().creid_center_loss_fwd(L.ptr(x), L.ptr(labels), L.ptr(centers), B, C, D, L.ptr(row),
                                              L.ptr(loss), L.stream()), "creid_center_loss_fwd")
        ctx.save_for_backward(x, labels, centers, row)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        x, labels, centers, row = ctx.saved_tensors
        B, D = x.shape
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dc = torch.zeros_like(centers) if ctx.needs_input_grad[2] else None
        g = _f32c(gloss.reshape(1))
        L.check(L.lib().creid_center_loss_bwd(L.ptr(x), L.ptr(labels), L.ptr(centers), L.ptr(row), B, D, L.ptr(g),
                                              1.0, L.ptr(dx), L.ptr(dc), L.stream()), "creid_center_loss_bwd")
        return dx, None, dc


# --------------------------------------------------------------------------- xent
class XentLabelSmoothFn(torch.autograd.Function):
    """losses/triplet_loss.py:194-205.  The gradient is produced in the forward pass (one kernel)."""

    @staticmethod
    def forward(ctx, logits, targets, eps):
        logits = _f32c(logits)
        targets = targets.to(torch.int64).contiguous()
        L.require_gpu(logits, targets)
        B, C = logits.shape
        row = torch.empty(B, dtype=torch.float32, device=logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlog = torch.empty_like(logits)
        L.check(L.lib().creid_xent_ls(L.ptr(logits), L.ptr(targets), B, C, float(eps), 1.0, L.ptr(row), L.ptr(loss),
                                      L.ptr(dlog), L.stream()), "creid_xent_ls")
        ctx.save_for_backward(dlog)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        (dlog,) = ctx.saved_tensors
        return dlog * gloss, None, None


# --------------------------------------------------------------------------- BNNeck
class BatchNorm1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rmean, rvar, training, momentum, eps):
        x = _f32c(x)
        L.require_gpu(x, weight, bias, rmean, rvar)
        B, D = x.shape
        y = torch.empty_like(x)
        sm = torch.empty(D, dtype=torch.float32, device=x.device) if training else None
        si = torch.empty(D, dtype=torch.float32, device=x.device) if training else None
        L.check(L.lib().creid_bn1d_fwd(L.ptr(x), B, D, L.ptr(weight), L.ptr(bias), L.ptr(rmean), L.ptr(rvar),
                                       1 if training else 0, float(momentum), float(eps), L.ptr(y), L.ptr(sm),
                                       L.ptr(si), L.stream()), "creid_bn1d_fwd")
        if training:
            ctx.save_for_backward(x, weight, sm, si)
        else:
            ctx.save_for_backward(x, weight, rmean.clone(), torch.rsqrt(rvar + eps))
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, sm, si = ctx.saved_tensors
        B, D = x.shape
        dy = _f32c(dy)
        if not ctx.training:
            return dy * (weight * si), None, None, None, None, None, None, None
        dx = torch.zeros_like(x)
        dw = torch.zeros_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.zeros_like(weight) if ctx.needs_input_grad[2] else None
        L.check(L.lib().creid_bn1d_bwd(L.ptr(x), L.ptr(dy), B, D, L.ptr(weight), L.ptr(sm), L.ptr(si), L.ptr(dx),
                                       L.ptr(dw), L.ptr(db), L.stream()), "creid_bn1d_bwd")
        return dx, dw, db, None, None, None, None, None


# --------------------------------------------------------------------------- classifier
def gemm_f32(A, sam, sak, Bm, sbk, sbn, M, N, K, out=None, alpha=1.0, beta=0.0, split_k=1):
    out = torch.empty((M, N), dtype=torch.float32, device=A.device) if out is None else out
    L.check(L.lib().creid_gemm_f32(L.ptr(A), sam, sak, L.ptr(Bm), sbk, sbn, L.ptr(out), N, M, N, K, float(alpha),
                                   float(beta), split_k, L.stream()), "creid_gemm_f32")
    return out


class LinearNoBiasFn(torch.autograd.Function):
    """y = x @ W^T  (fc_query, modelling/bases.py:86; W is [C, D])."""

    @staticmethod
    def forward(ctx, x, w):
        x = _f32c(x); w = _f32c(w)
        L.require_gpu(x, w)
        B, D = x.shape
        C = w.shape[0]
        ctx.save_for_backward(x, w)
        # few output tiles, long K: split K 8-way (partials combined by fp32 atomics; order-dependent in the last
        # bits only -- set CREID_DETERMINISTIC=1 for the single-pass kernel)
        sk = 1 if _DETERMINISTIC else 8
        return gemm_f32(x, D, 1, w, 1, D, B, C, D, split_k=sk)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        B, D = x.shape
        C = w.shape[0]
        dx = gemm_f32(dy, C, 1, w, D, 1, B, D, C, split_k=1 if _DETERMINISTIC else 4) if ctx.needs_input_grad[0] else None  # dy @ W
        dw = gemm_f32(dy, 1, C, x, D, 1, C, D, B) if ctx.needs_input_grad[1] else None      # dy^T @ x
        return dx, dw
