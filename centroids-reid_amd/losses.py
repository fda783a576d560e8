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


def euclidean_dist(x, y=None):
    """losses/triplet_loss.py:27-41 for the x-vs-x case used by the loss."""
    if y is not None and y is not x:
        raise NotImplementedError("only the self-distance used by TripletLoss is on the hot path")
    labels = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
    return ops.pairwise_dist_mine(x, labels)[0]


def hard_example_mining(features, labels, return_inds=False):
    """losses/triplet_loss.py:68-119, fused with the distance computation (takes FEATURES)."""
    _, dap, dan, pi, ni = ops.pairwise_dist_mine(features, labels)
    return (dap, dan, pi, ni) if return_inds else (dap, dan)


class TripletLoss(object):
    """losses/triplet_loss.py:122-173."""

    def __init__(self, margin=None, dist_func="euclidean"):
        self.margin = margin
        if dist_func != "euclidean":
            raise NotImplementedError("SOLVER.DISTANCE_FUNC='cosine' is not on the accelerated path yet")
        self.dist_func = dist_func

    def __call__(self, global_feat, labels, warmup_margin=False, print_data=False, normalize_feature=False,
                 mask=None):
        if normalize_feature:
            raise NotImplementedError("normalize_feature=True is never used by the reference's training step")
        loss, dist_ap, dist_an, _ = ops.TripletHardMine.apply(global_feat, labels, mask, self.margin)
        if mask is not None:                      # losses/triplet_loss.py:148-151
            dist_ap, dist_an = dist_ap[mask], dist_an[mask]
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
