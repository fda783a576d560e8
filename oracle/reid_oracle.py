"""CPU oracle for the centroids-reid embedding-and-retrieval hot path (heads + eval).

TEST INFRASTRUCTURE ONLY.  This file is a clean-room CPU restatement (torch-CPU
fp32 / numpy) of the reference algorithm; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker /
reported baseline.  The product package (``centroids-reid_amd``) never imports it
and fails loudly when the HIP library is missing.

Pinning: the upstream repository ships NO tests, fixtures or golden vectors
(SURVEY.md §4), so parity is pinned by outputs of the reference itself, imported
in the build container by ``tools/gen_golden.py`` (under ``tools/ref_import.py``
stubs) and committed as data under ``tests/golden/``.  ``tests/test_oracle_golden.py``
checks every function here against those vectors.

Every function cites the reference file:line (relative to the upstream repo root)
whose arithmetic it restates.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

K_LIST = (1, 5, 10, 20, 50)  # utils/eval_reid.py:15


# ----------------------------------------------------------------------------
# Stage D: embeddings -> distance matrix
# ----------------------------------------------------------------------------
def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """utils/reid_metric.py:113-115 -> F.normalize(p=2, dim=1): x / max(||x||_2, 1e-12)."""
    n = torch.sqrt((x * x).sum(dim=1, keepdim=True))
    return x / torch.clamp(n, min=1e-12)


def sqdist_matrix(q: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """utils/reid_metric.py:25-33 get_euclidean: SQUARED L2, no clamp, no sqrt:
    d[i,j] = (|q_i|^2 + |g_j|^2) + (-2) * <q_i, g_j>."""
    qq = (q * q).sum(dim=1, keepdim=True)
    gg = (g * g).sum(dim=1, keepdim=True).t()
    return (qq + gg) - 2.0 * (q @ g.t())


def cosine_dist_matrix(q: torch.Tensor, g: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils/reid_metric.py:37-59 get_cosine: clamp(|1 - cos|, min=eps)."""
    qn = q / torch.clamp(q.norm(dim=1, keepdim=True), min=eps)
    gn = g / torch.clamp(g.norm(dim=1, keepdim=True), min=eps)
    return torch.abs(1 - qn @ gn.t()).clamp(min=eps)


def rank_rows(dist) -> np.ndarray:
    """utils/reid_metric.py:129,132 np.argsort(distmat, axis=1).  The reference's sort
    kind is unstable (ties undefined); the build DEFINES the order
    as ascending (distance, gallery index), i.e. a stable sort."""
    d = dist.numpy() if isinstance(dist, torch.Tensor) else np.asarray(dist)
    return np.argsort(d, axis=1, kind="stable").astype(np.int64)


# ----------------------------------------------------------------------------
# Stage E: CMC / mAP / top-k
# ----------------------------------------------------------------------------
def eval_market(indices, q_pids, g_pids, q_camids, g_camids, max_rank: int = 50, chunk: int = 256):
    """utils/eval_reid.py:25-92 eval_func (respect_camids=False), vectorised over queries.

    Returns (all_cmc float32[max_rank], mAP float64, all_topk float64[5],
             per_query dict(valid bool[nq], ap float64[nq], first int64[nq])).
    per-query `first` is the 0-based kept-rank of the first correct match (-1 if invalid).
    """
    indices = np.asarray(indices)
    q_pids = np.asarray(q_pids); g_pids = np.asarray(g_pids)
    q_camids = np.asarray(q_camids); g_camids = np.asarray(g_camids)
    nq, ng = indices.shape
    if ng < max_rank:  # eval_reid.py:33-35
        max_rank = ng
    valid = np.zeros(nq, bool)
    ap = np.zeros(nq, np.float64)
    first = np.full(nq, -1, np.int64)
    for s in range(0, nq, chunk):
        idx = indices[s:s + chunk]
        gp = g_pids[idx]
        match = gp == q_pids[s:s + chunk, None]                       # :36
        remove = match & (g_camids[idx] == q_camids[s:s + chunk, None])  # :57
        keep = ~remove
        mk = match & keep                                             # orig_cmc (:62) in place
        v = mk.any(axis=1)                                            # :63-65
        kpos = np.cumsum(keep, axis=1)                                # 1-based kept position
        cum = np.cumsum(mk, axis=1)                                   # :75
        contrib = np.where(mk, cum / np.maximum(kpos, 1).astype(np.float64), 0.0)  # :76-77
        nrel = mk.sum(axis=1)
        a = contrib.sum(axis=1) / np.maximum(nrel, 1)                 # :78
        f = np.where(v, kpos[np.arange(len(idx)), mk.argmax(axis=1)] - 1, -1)
        valid[s:s + chunk] = v; ap[s:s + chunk] = np.where(v, a, 0.0); first[s:s + chunk] = f
    nvalid = float(valid.sum())
    ranks = np.arange(max_rank)[None, :]
    cmc_rows = (first[valid][:, None] <= ranks).astype(np.float32)   # :67-70 clipped cumsum
    all_cmc = (cmc_rows.sum(0) / nvalid).astype(np.float32)          # :86-87
    mAP = float(np.mean(ap[valid]))                                  # :88
    topk = np.stack([(first[valid] < k) for k in K_LIST], axis=1).astype(np.int64)  # :18-22
    all_topk = topk.mean(axis=0)                                     # :89-90
    return all_cmc, mAP, all_topk, dict(valid=valid, ap=ap, first=first)


def val_centroids(emb: torch.Tensor, labels, camids, num_query: int):
    """modelling/bases.py:179-262 validation_create_centroids (respect_camids=False):
    gallery -> per-PID mean in sorted-unique-PID order; query rows kept; dummy camids 0/1."""
    labels = np.asarray(labels)
    eq, lq = emb[:num_query], labels[:num_query]
    eg, lg = emb[num_query:], labels[num_query:]
    uniq = np.unique(lg)  # sorted (:200)
    cents = torch.stack([eg[torch.from_numpy(np.nonzero(lg == u)[0])].sum(0) / int((lg == u).sum())
                         for u in uniq])  # :92-95, :238-241
    out = torch.cat([eq, cents], 0)
    out_labels = np.hstack([lq, uniq])
    # :255-260 quirk: ones_like() is taken of the ALREADY query-prefixed label vector, so the
    # returned camid vector has nq + (nq + n_centroids) entries (tail of ones; harmless
    # because only g_camids[:n_centroids] is ever indexed downstream).
    out_cam = np.hstack([np.zeros_like(lq), np.ones_like(out_labels)])
    return out, out_labels, out_cam


def val_centroids_camera(emb: torch.Tensor, labels, camids, num_query: int):
    """modelling/bases.py:179-262 with respect_camids=True.  For every gallery PID and every distinct camera of
    its QUERIES, one centroid of the gallery rows seen by OTHER cameras, de-duplicated by camera set.  Quirk
    kept on purpose (:214): the gallery rows' camera ids are looked up as camids[inds] with GALLERY-relative
    indices on the FULL (query-first) camid vector.  Returns (emb [nq + n_cent, D], labels, cam_sets) where
    cam_sets is a list of lists ([cam] for each query, the used-camera list for each centroid)."""
    labels = np.asarray(labels); camids = np.asarray(camids)
    eq, lq = emb[:num_query], labels[:num_query]
    eg, lg = emb[num_query:], labels[num_query:]
    cents, cl, cc = [], [], []
    for u in np.unique(lg):
        inds = np.nonzero(lg == u)[0]
        inds_q = np.nonzero(lq == u)[0]
        cams_g = camids[inds]                      # the reference's indexing quirk
        seen = set()
        for cur in np.unique(camids[inds_q]):
            sel = np.nonzero(cams_g != cur)[0]
            if len(sel) == 0:
                continue
            used = tuple(sorted(np.unique(cams_g[cams_g != cur]).tolist()))
            if used in seen:
                continue
            seen.add(used)
            rows = eg[torch.from_numpy(inds[sel])]
            cents.append(rows.sum(0) / rows.shape[0]); cl.append(u); cc.append(list(used))
    out = torch.cat([eq, torch.stack(cents)], 0)
    return out, np.hstack([lq, np.asarray(cl)]), [[int(c)] for c in camids[:num_query]] + cc


def eval_market_camsets(indices, q_pids, g_pids, q_cams, g_cam_sets, max_rank: int = 50):
    """utils/eval_reid.py:25-92 with respect_camids=True (:51-55): a gallery entry is dropped for a query iff it
    has the query's pid AND the query's camera is in the entry's camera SET."""
    indices = np.asarray(indices)
    nq, ng = indices.shape
    member = np.zeros((ng, 64), bool)
    for j, cs in enumerate(g_cam_sets):
        member[j, list(cs)] = True
    q_pids = np.asarray(q_pids); g_pids = np.asarray(g_pids); q_cams = np.asarray(q_cams)
    if ng < max_rank:
        max_rank = ng
    valid = np.zeros(nq, bool); ap = np.zeros(nq); first = np.full(nq, -1, np.int64)
    for qi in range(nq):
        order = indices[qi]
        match = g_pids[order] == q_pids[qi]
        keep = ~(match & member[order, q_cams[qi]])
        mk = match & keep
        if not mk.any():
            continue
        kpos = np.cumsum(keep); cum = np.cumsum(mk)
        valid[qi] = True
        ap[qi] = (np.where(mk, cum / np.maximum(kpos, 1), 0.0)).sum() / mk.sum()
        first[qi] = kpos[mk.argmax()] - 1
    nv = float(valid.sum())
    cmc = ((first[valid][:, None] <= np.arange(max_rank)[None, :]).astype(np.float32).sum(0) / nv).astype(np.float32)
    topk = np.stack([(first[valid] < k) for k in K_LIST], axis=1).astype(np.int64).mean(0)
    return cmc, float(np.mean(ap[valid])), topk, dict(valid=valid, ap=ap, first=first)


def r1_map(feats: torch.Tensor, pids, camids, num_query: int, feat_norm: bool = True,
           dist: str = "euclidean"):
    """utils/reid_metric.py:112-151 R1_mAP.compute."""
    feats = feats.float()
    if feat_norm:
        feats = l2_normalize(feats)
    q, g = feats[:num_query], feats[num_query:]
    d = sqdist_matrix(q, g) if dist == "euclidean" else cosine_dist_matrix(q, g)
    idx = rank_rows(d)
    pids = np.asarray(pids); camids = np.asarray(camids)
    cmc, mAP, topk, per_q = eval_market(idx, pids[:num_query], pids[num_query:],
                                        camids[:num_query], camids[num_query:])
    return cmc, mAP, topk, dict(dist=d, indices=idx, **per_q)


# ----------------------------------------------------------------------------
# Stage B: masks + leave-one-out centroids
# ----------------------------------------------------------------------------
def create_masks_train(labels) -> tuple[np.ndarray, list]:
    """modelling/bases.py:359-384: masks[i, j] == False iff j is the i-th occurrence of its
    PID (batch order); a PID with fewer than max-count occurrences gets its whole
    [start,end) block (cumulative-count addressing, :379-382) cleared in the extra rounds."""
    labels = np.asarray(labels)
    order = {}
    for j, p in enumerate(labels.tolist()):
        order.setdefault(p, []).append(j)
    groups = list(order.values())
    lens = [len(g) for g in groups]
    cs = np.cumsum(lens)
    kmax = max(lens)
    masks = np.ones((kmax, len(labels)), bool)
    for i in range(kmax):
        for gi, g in enumerate(groups):
            if i < len(g):
                masks[i, g[i]] = False
            else:
                start = cs[gi - 1]  # NB gi==0 -> cs[-1] (reference quirk, :380)
                masks[i, start:start + lens[gi]] = False
    return masks, groups


def loo_centroids(features: torch.Tensor, is_real: torch.Tensor, P: int, K: int):
    """train_ctl_model.py:79-104.  Batch is PID-contiguous [P,K].  Round i holds out slot i:
    centroid[i,p] = sum_{s != i, real} f[p,s] / max(count,1) if slot i of p is real else 0.
    Returns (centroids [K,P,D], valid_inst int64 [K,P])."""
    D = features.shape[1]
    f = features.view(P, K, D)
    ir = is_real.view(P, K)
    cents, valid = [], []
    for i in range(K):
        g = ir.clone()
        g[:, i] = False
        g = g & ir[:, i:i + 1]
        cnt = g.sum(1)
        acc = torch.zeros(P, D, dtype=features.dtype)
        for s in range(K):  # same s-order as the reference's sum(-2)
            acc = acc + g[:, s:s + 1].to(features.dtype) * f[:, s]
        cents.append(acc / cnt.clamp(min=1).unsqueeze(-1).to(features.dtype))
        valid.append(cnt)
    return torch.stack(cents), torch.stack(valid)


# ----------------------------------------------------------------------------
# Stage C: losses
# ----------------------------------------------------------------------------
def euclidean_dist(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """losses/triplet_loss.py:27-41: sqrt(clamp(|x|^2 + |y|^2 - 2 x y^T, min=1e-12))."""
    xx = (x * x).sum(1, keepdim=True)
    yy = (y * y).sum(1, keepdim=True).t()
    d = (xx + yy) - 2.0 * (x.float() @ y.float().t())
    return d.clamp(min=1e-12).sqrt()


def hard_example_mining(dist: torch.Tensor, labels: torch.Tensor):
    """losses/triplet_loss.py:68-119: per anchor hardest positive (max, includes self) and
    hardest negative (min); first index wins ties (torch.max/min on the compacted row)."""
    same = labels.view(-1, 1) == labels.view(1, -1)
    big = torch.finfo(dist.dtype).max
    d_ap, p_idx = torch.where(same, dist, torch.full_like(dist, -big)).max(dim=1)
    d_an, n_idx = torch.where(same, torch.full_like(dist, big), dist).min(dim=1)
    return d_ap, d_an, p_idx, n_idx


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """losses/triplet_loss.py:16-24 `normalize`: x / (|x|_2 + 1e-12) (NOT F.normalize: the epsilon is added)."""
    return x / (torch.sqrt((x * x).sum(dim=-1, keepdim=True)) + 1e-12)


def cosine_dist(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """losses/triplet_loss.py:44-65: rows / max(|row|, eps), similarity matrix, |1 - sim| clamped at eps."""
    xn = x / torch.clamp(torch.sqrt((x * x).sum(1, keepdim=True)), min=eps)
    yn = y / torch.clamp(torch.sqrt((y * y).sum(1, keepdim=True)), min=eps)
    return torch.abs(1 - xn @ yn.t()).clamp(min=eps)


def triplet_loss(feat: torch.Tensor, labels: torch.Tensor, margin=0.5, mask=None, dist="euclidean",
                 normalize_feature=False):
    """losses/triplet_loss.py:139-173 TripletLoss.__call__.  `mask` filters anchors AFTER mining (:148-151).
    margin None -> SoftMarginLoss(an - ap, 1).  dist = SOLVER.DISTANCE_FUNC (:134-137); normalize_feature
    (:141-142) rescales the rows first."""
    if normalize_feature:
        feat = normalize_rows(feat)
    d = cosine_dist(feat, feat) if dist == "cosine" else euclidean_dist(feat, feat)
    d_ap, d_an, _, _ = hard_example_mining(d, labels)
    if mask is not None:
        d_ap, d_an = d_ap[mask], d_an[mask]
    if margin is not None:
        loss = torch.clamp(d_ap - d_an + margin, min=0).mean()  # MarginRankingLoss(an, ap, 1)
    else:
        loss = F.softplus(-(d_an - d_ap)).mean()                 # SoftMarginLoss
    return loss, d_ap, d_an


def center_loss(x: torch.Tensor, labels: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    """losses/center_loss.py:26-46: sum_b clamp(|x_b - c_{y_b}|^2, 1e-12, 1e12)/B computed via
    the expanded form, PLUS (C-1)*1e-12 from clamping the B*(C-1) masked-out zeros."""
    B, C = x.shape[0], centers.shape[0]
    d = (x * x).sum(1, keepdim=True) + (centers * centers).sum(1, keepdim=True).t() \
        - 2.0 * (x.float() @ centers.t())
    own = d.gather(1, labels.view(-1, 1)).squeeze(1).clamp(min=1e-12, max=1e12)
    return (own.sum() + B * (C - 1) * 1e-12) / B


def xent_label_smooth(logits: torch.Tensor, targets: torch.Tensor, eps: float = 0.1) -> torch.Tensor:
    """losses/triplet_loss.py:194-205: (-t * log_softmax).mean(0).sum(), t = (1-eps)*onehot + eps/C."""
    C = logits.shape[1]
    lp = F.log_softmax(logits, dim=1)
    t = torch.full_like(lp, eps / C)
    t.scatter_(1, targets.view(-1, 1), 1 - eps + eps / C)
    return (-t * lp).mean(0).sum()


def ctl_heads(features, labels, is_real, bn_weight, bn_bias, bn_rm, bn_rv, fc_weight, centers,
              P, K, margin=0.5, center_w=5e-4, xent_w=1.0, query_w=1.0, centroid_w=1.0):
    """train_ctl_model.py:59-152 (everything after the backbone, before backward).
    Returns dict of torch scalars (autograd-capable) + stats."""
    out = {}
    lq, _, _ = triplet_loss(features, labels, margin, mask=is_real)       # :62-67
    out["query_triplet"] = lq * query_w
    fr, lr = features[is_real], labels[is_real]                            # :69-70
    out["query_center"] = center_w * center_loss(fr, lr, centers)          # :71-73
    bnf = F.batch_norm(fr, bn_rm, bn_rv, bn_weight, bn_bias, True, 0.1, 1e-5)  # :74
    out["query_xent"] = xent_label_smooth(bnf @ fc_weight.t(), lr) * xent_w    # :75-77
    cents, valid = loo_centroids(features, is_real, P, K)                  # :79-104
    ir = is_real.view(P, K)
    losses, aps, ans, norms = [], [], [], []
    for i in range(K):
        if int((valid[i] > 0).sum()) <= 1:                                 # :113-114
            continue
        qsel = ir[:, i]                                                    # ~mask[i] & t_re[i]
        qf = features.view(P, K, -1)[:, i][qsel]
        ql = labels.view(P, K)[:, i][qsel]
        c = cents[i]
        c = c[c.abs().sum(1) > 1e-7]                                       # :120-122
        emb = torch.cat([qf, c]); lab = torch.cat([ql, ql])                # :123-124
        if emb.shape[0] != lab.shape[0]:
            raise RuntimeError("query/centroid count mismatch (reference raises in expand)")
        l, dap, dan = triplet_loss(emb, lab, margin)
        losses.append(l); aps.append(dap.detach().mean()); ans.append(dan.detach().mean())
        norms.append(c.norm(dim=1).mean())                                 # :138-139
    out["centroid_triplet"] = torch.stack(losses).mean() * centroid_w      # :142-145
    out["step_dist_ap"] = torch.stack(aps).mean()
    out["step_dist_an"] = torch.stack(ans).mean()
    out["l2_mean_centroid"] = torch.stack(norms).mean().detach()
    out["total"] = (out["centroid_triplet"] + out["query_center"] + out["query_xent"]
                    + out["query_triplet"])                                # :150-152
    return out


def center_sgd_step(centers, grad, center_w=5e-4, lr=0.5):
    """train_ctl_model.py:157-159 + solver/build.py:44: grad *= 1/center_w; SGD(lr) step."""
    return centers - lr * (grad * (1.0 / center_w))


def adam_step(p, g, m, v, step, lr, wd=5e-4, b1=0.9, b2=0.999, eps=1e-8):
    """solver/build.py:36-39 torch.optim.Adam (L2 weight decay folded into the gradient,
    bias-corrected, eps added after sqrt(v_hat)) -- restated from the published algorithm."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)) + eps
    return p - (lr / bc1) * (m / denom), m, v
