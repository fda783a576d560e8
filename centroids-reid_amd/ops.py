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
    def forward(ctx, x, labels, mask, margin, cosine=False):
        """cosine=True: `x` holds unit-length rows (RowNormalize mode 0) and the distance is
        clamp(|1 - x_i.x_j|, 1e-12) (losses/triplet_loss.py:57-65)."""
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
        fn = L.lib().creid_triplet_cosine_fwd if cosine else L.lib().creid_triplet_fwd
        L.check(fn(L.ptr(x), L.ptr(labels), L.ptr(m8), N, D, float(margin) if margin is not None else -1.0, L.ptr(dap),
                   L.ptr(dan), L.ptr(pi), L.ptr(ni), L.ptr(coef), L.ptr(out4), None, L.stream()), "creid_triplet_fwd")
        ctx.save_for_backward(x, dap, dan, pi, ni, coef)
        ctx.cosine = cosine
        ctx.mark_non_differentiable(dap, dan, out4)
        return out4[0].clone(), dap, dan, out4

    @staticmethod
    def backward(ctx, gloss, _gap, _gan, _g4):
        x, dap, dan, pi, ni, coef = ctx.saved_tensors
        N, D = x.shape
        dx = torch.zeros_like(x)
        g = _f32c(gloss.reshape(1))
        fn = L.lib().creid_triplet_cosine_bwd if ctx.cosine else L.lib().creid_triplet_bwd
        L.check(fn(L.ptr(x), N, D, L.ptr(dap), L.ptr(dan), L.ptr(pi), L.ptr(ni), L.ptr(coef), L.ptr(g), 1.0, L.ptr(dx),
                   L.stream()), "creid_triplet_bwd")
        return dx, None, None, None, None


class RowNormalize(torch.autograd.Function):
    """mode 0: x / max(|x|, eps) (cosine_similarity, losses/triplet_loss.py:50-52);
    mode 1: x / (|x| + eps) (`normalize`, losses/triplet_loss.py:16-24)."""

    @staticmethod
    def forward(ctx, x, mode, eps):
        x = _f32c(x)
        L.require_gpu(x)
        N, D = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(N, dtype=torch.float32, device=x.device)
        L.check(L.lib().creid_rownorm_fwd(L.ptr(x), N, D, int(mode), float(eps), L.ptr(y), L.ptr(norm), L.stream()),
                "creid_rownorm_fwd")
        ctx.save_for_backward(x, norm)
        ctx.cfg = (int(mode), float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, norm = ctx.saved_tensors
        mode, eps = ctx.cfg
        N, D = x.shape
        dx = torch.empty_like(x)
        L.check(L.lib().creid_rownorm_bwd(L.ptr(x), L.ptr(norm), L.ptr(_f32c(dy)), N, D, mode, eps, L.ptr(dx), L.stream()),
                "creid_rownorm_bwd")
        return dx, None, None


class HardMineFromDist(torch.autograd.Function):
    """hard_example_mining(dist_mat, labels, return_inds=True) (losses/triplet_loss.py:68-119); the gradient
    flows back to the two selected entries of each row, like the reference's masked max / min."""

    @staticmethod
    def forward(ctx, dist_mat, labels):
        d = _f32c(dist_mat)
        labels = labels.to(torch.int64).contiguous()
        L.require_gpu(d, labels)
        N = d.shape[0]
        dev = d.device
        dap = torch.empty(N, dtype=torch.float32, device=dev); dan = torch.empty_like(dap)
        pi = torch.empty(N, dtype=torch.int32, device=dev); ni = torch.empty_like(pi)
        L.check(L.lib().creid_hard_mine_from_dist(L.ptr(d), L.ptr(labels), N, L.ptr(dap), L.ptr(dan), L.ptr(pi), L.ptr(ni),
                                                  L.stream()), "creid_hard_mine_from_dist")
        pi64, ni64 = pi.long(), ni.long()
        ctx.save_for_backward(pi64, ni64)
        ctx.N = N
        ctx.mark_non_differentiable(pi64, ni64)
        return dap, dan, pi64, ni64

    @staticmethod
    def backward(ctx, gap, gan, _gp, _gn):
        pi, ni = ctx.saved_tensors
        g = torch.zeros((ctx.N, ctx.N), dtype=torch.float32, device=pi.device)
        g.scatter_add_(1, pi[:, None], gap[:, None].float())
        g.scatter_add_(1, ni[:, None], gan[:, None].float())
        return g, None


class EuclideanDist(torch.autograd.Function):
    """euclidean_dist(x, y) of losses/triplet_loss.py:27-41 for two different row sets: the squared matrix comes
    from the MFMA distance kernel of the evaluation stage, then clamp(min=1e-12).sqrt() in place."""

    @staticmethod
    def forward(ctx, x, y):
        x = _f32c(x); y = _f32c(y)
        L.require_gpu(x, y)
        lib = L.lib()
        m, n, D = x.shape[0], y.shape[0], x.shape[1]
        dev = x.device
        xx = torch.empty(m, dtype=torch.float32, device=dev); yy = torch.empty(n, dtype=torch.float32, device=dev)
        L.check(lib.creid_row_sqnorm(L.ptr(x), L.ptr(xx), m, D, L.F32, L.stream()), "creid_row_sqnorm")
        L.check(lib.creid_row_sqnorm(L.ptr(y), L.ptr(yy), n, D, L.F32, L.stream()), "creid_row_sqnorm")
        d = torch.empty((m, n), dtype=torch.float32, device=dev)
        L.check(lib.creid_sqdist_matrix(L.ptr(x), L.ptr(y), L.ptr(xx), L.ptr(yy), m, n, D, L.F32, L.ptr(d), n, L.stream()),
                "creid_sqdist_matrix")
        L.check(lib.creid_clamp_sqrt_inplace(L.ptr(d), m * n, 1e-12, L.stream()), "creid_clamp_sqrt_inplace")
        ctx.save_for_backward(x, y, d)
        return d

    @staticmethod
    def backward(ctx, g):
        x, y, d = ctx.saved_tensors
        m, n, D = x.shape[0], y.shape[0], x.shape[1]
        # d dist_ij / d x_i = (x_i - y_j) / dist_ij, zero where the clamp was active (dist == 1e-6)
        w = torch.where(d > 1e-6, _f32c(g) / d, torch.zeros_like(d)).contiguous()
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = w.sum(1, keepdim=True) * x - gemm_f32(w, n, 1, y, D, 1, m, D, n)          # rowsum(W) x - W @ y
        if ctx.needs_input_grad[1]:
            dy = w.sum(0)[:, None] * y - gemm_f32(w, 1, n, x, D, 1, n, D, m)               # colsum(W) y - W^T @ x
        return dx, dy


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
        elif any(ctx.needs_input_grad[:3]):  # eval mode under autograd only: inference (no_grad) must not pay three extra launches
            ctx.save_for_backward(x, weight, rmean.clone(), torch.rsqrt(rvar + eps))
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, sm, si = ctx.saved_tensors
        B, D = x.shape
        dy = _f32c(dy)
        if not ctx.training:
            # running statistics are constants: y = (x - rm) * w * si + b.  x may be detached while weight / bias still want
            # their gradient (eval-mode BNNeck on detached engine features), so every one of the three is optional
            dx = dy * (weight * si) if ctx.needs_input_grad[0] else None
            dw = (dy * (x - sm) * si).sum(0) if ctx.needs_input_grad[1] else None
            db = dy.sum(0) if ctx.needs_input_grad[2] else None
            return dx, dw, db, None, None, None, None, None
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
