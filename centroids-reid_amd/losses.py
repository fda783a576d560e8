"""Loss surface of the reference (losses/triplet_loss.py, losses/center_loss.py) on the HIP kernels.

Same class names, constructor arguments and call signatures as the reference:
  TripletLoss(margin, dist_func)(feat, labels, ..., mask=None) -> (loss, dist_ap, dist_an)
  CenterLoss(num_classes, feat_dim).forward(x, labels) -> loss          (.centers is the Parameter)
  CrossEntropyLabelSmooth(num_classes, epsilon).forward(logits, targets) -> loss
plus euclidean_dist / hard_example_mining helpers.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


def normalize(x, axis=-1):
    """losses/triplet_loss.py:16-24: x / (|x|_2 + 1e-12) along the last dimension of a [N, D] tensor."""
    assert x.dim() == 2 and axis in (-1, 1)
    return ops.RowNormalize.apply(x, 1, 1e-12)


def euclidean_dist(x, y):
    """losses/triplet_loss.py:27-41: sqrt(clamp(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 1e-12)) -> [m, n]."""
    return ops.EuclideanDist.apply(x, y)


def cosine_similarity(x, y, eps=1e-12):
    """losses/triplet_loss.py:44-54 (rows scaled by 1 / max(|.|, eps), then the inner-product matrix)."""
    xn, yn = ops.RowNormalize.apply(x, 0, eps), ops.RowNormalize.apply(y, 0, eps)
    return ops.LinearNoBiasFn.apply(xn, yn)                  # xn @ yn^T on the HIP GEMM


def cosine_dist(x, y, eps=1e-12):
    """losses/triplet_loss.py:57-65."""
    return torch.abs(1 - cosine_similarity(x, y, eps)).clamp(min=eps)


def hard_example_mining(dist_mat, labels, return_inds=False):
    """losses/triplet_loss.py:68-119 on a square distance matrix [N, N]."""
    assert dist_mat.dim() == 2 and dist_mat.shape[0] == dist_mat.shape[1]
    dap, dan, pi, ni = ops.HardMineFromDist.apply(dist_mat, labels)
    return (dap, dan, pi, ni) if return_inds else (dap, dan)


class TripletLoss(object):
    """losses/triplet_loss.py:122-173.  The distance matrix, the mining and the loss run as ONE fused pass over
    the features (the [N, N] matrix is never written); `dist_func` selects the euclidean or the cosine form and
    normalize_feature=True prepends the reference's `normalize` (:141-142)."""

    def __init__(self, margin=None, dist_func="euclidean"):
        self.margin = margin
        if dist_func not in ("euclidean", "cosine"):
            raise KeyError(dist_func)             # the reference leaves self.dist_func unset -> AttributeError on call
        self.dist_name = dist_func
        self.dist_func = cosine_dist if dist_func == "cosine" else euclidean_dist

    def __call__(self, global_feat, labels, warmup_margin=False, print_data=False, normalize_feature=False,
                 mask=None):
        if normalize_feature:
            global_feat = normalize(global_feat, axis=-1)
        if self.dist_name == "cosine":
            unit = ops.RowNormalize.apply(global_feat, 0, 1e-12)
            loss, dist_ap, dist_an, _ = ops.TripletHardMine.apply(unit, labels, mask, self.margin, True)
        else:
            loss, dist_ap, dist_an, _ = ops.TripletHardMine.apply(global_feat, labels, mask, self.margin)
        if mask is not None:                      # losses/triplet_loss.py:148-151
            dist_ap, dist_an = dist_ap[mask], dist_an[mask]
        if print_data:                            # :158-171
            print(f"LOSS: {loss.item()}")
            print(f"precision: {(dist_an > dist_ap).float().mean()}")
            if self.margin is not None:
                print(f"proportion of triplets that satisfy margin: {(dist_an > dist_ap + self.margin).float().mean()}")
            print(f"AP mean distance: {dist_ap.mean()}")
            print(f"AN mean distance: {dist_an.mean()}")
        return loss, dist_ap, dist_an


class CrossEntropyLabelSmooth(nn.Module):
    """losses/triplet_loss.py:176-205."""

    def __init__(self, num_classes, epsilon=0.1, use_gpu=True):
        super().__init__()
        self.num_classes = num_classes
        self.epsilon = epsilon
        self.use_gpu = use_gpu

    def forward(self, inputs, targets):
        assert inputs.shape[1] == self.num_classes
        return ops.XentLabelSmoothFn.apply(inputs, targets, self.epsilon)


class CenterLoss(nn.Module):
    """losses/center_loss.py:4-46.  `centers` ~ randn [num_classes, feat_dim]."""

    def __init__(self, num_classes=751, feat_dim=2048, use_gpu=True):
        super().__init__()
        self.num_classes = num_classes
        self.feat_dim = feat_dim
        self.use_gpu = use_gpu
        self.centers = nn.Parameter(torch.randn(self.num_classes, self.feat_dim))

    def forward(self, x, labels):
        assert x.size(0) == labels.size(0), "features.size(0) is not equal to labels.size(0)"
        return ops.CenterLossFn.apply(x, labels, self.centers)


class BatchNorm1d(nn.Module):
    """nn.BatchNorm1d(d_model) of modelling/bases.py:83 (same parameter / buffer names)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        if self.training:
            self.num_batches_tracked += 1
        return ops.BatchNorm1dFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                       self.training, self.momentum, self.eps)


class Linear(nn.Module):
    """nn.Linear(d_model, num_classes, bias=False), init N(0, 0.001) (modelling/bases.py:29-34,86-87)."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        assert not bias
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * 0.001)

    def forward(self, x):
        return ops.LinearNoBiasFn.apply(x, self.weight)
