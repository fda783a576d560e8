"""Stage D/E host side: the reference's metric surface on top of the HIP kernels.

Mirrors utils/reid_metric.py (get_euclidean :25-33, get_dist_func :62-68, R1_mAP :71-151)
and utils/eval_reid.py (eval_func :25-92) -- same names, argument meaning and return
values -- but every step runs on the GPU: rows are normalised, the squared-L2 matrix comes
from the MFMA distance kernel, ranking is a device radix sort, and CMC/AP are scanned on the
device; only the final few scalars come back to the host.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib as L

K_LIST = [1, 5, 10, 20, 50]


# --------------------------------------------------------------------------- primitives
def l2_normalize(x: torch.Tensor, out_dtype=torch.float32, eps: float = 1e-12, return_sqnorm=False):
    L.require_gpu(x)
    assert x.dtype == torch.float32 and x.dim() == 2
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    sq = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if return_sqnorm else None
    L.check(L.lib().creid_l2norm_rows(L.ptr(x), L.ptr(y), L.ptr(sq), x.shape[0], x.shape[1],
                                      L._DT[out_dtype], eps, L.stream()), "creid_l2norm_rows")
    return (y, sq) if return_sqnorm else y


def row_sqnorm(x: torch.Tensor) -> torch.Tensor:
    L.require_gpu(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(L.lib().creid_row_sqnorm(L.ptr(x), L.ptr(out), x.shape[0], x.shape[1], L.dtype_code(x), L.stream()),
            "creid_row_sqnorm")
    return out


def get_euclidean(x: torch.Tensor, y: torch.Tensor, xx=None, yy=None, **kwargs) -> torch.Tensor:
    """Squared L2 matrix [m, n] fp32 (utils/reid_metric.py:25-33)."""
    L.require_gpu(x, y)
    assert x.dtype == y.dtype and x.shape[1] == y.shape[1]
    xx = row_sqnorm(x) if xx is None else xx
    yy = row_sqnorm(y) if yy is None else yy
    m, n, D = x.shape[0], y.shape[0], x.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    L.check(L.lib().creid_sqdist_matrix(L.ptr(x), L.ptr(y), L.ptr(xx), L.ptr(yy), m, n, D, L.dtype_code(x),
                                        L.ptr(out), n, L.stream()), "creid_sqdist_matrix")
    return out


def get_cosine(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils/reid_metric.py:51-59: clamp(|1 - cos|, eps).  With unit rows |x-y|^2 = 2 - 2cos, so the
    cosine matrix is derived from the same MFMA kernel: cos = 1 - d/2 on re-normalised rows."""
    xn, xs = l2_normalize(x.float(), eps=eps, return_sqnorm=True)
    yn, ys = l2_normalize(y.float(), eps=eps, return_sqnorm=True)
    d = get_euclidean(xn, yn, xs, ys)
    cos = (xs[:, None] + ys[None, :] - d) * 0.5
    return torch.abs(1 - cos).clamp(min=eps)


def _pad_width(feats: torch.Tensor) -> torch.Tensor:
    """The kernels load 16-byte k-chunks (D % 4 == 0).  Zero columns change neither a norm nor a dot product, so any other
    width is padded up -- results identical, no kernel special case."""
    D = feats.shape[1]
    if D % 4 == 0:
        return feats
    return torch.nn.functional.pad(feats, (0, 4 - D % 4)).contiguous()


def get_dist_func(func_name="euclidean"):
    if func_name == "cosine":
        return get_cosine
    if func_name == "euclidean":
        return get_euclidean
    raise KeyError(func_name)


def rank_rows(distmat: torch.Tensor) -> torch.Tensor:
    """np.argsort(distmat, axis=1) (utils/reid_metric.py:129,132) with (distance, index) order."""
    L.require_gpu(distmat)
    assert distmat.dtype == torch.float32 and distmat.dim() == 2
    m, n = distmat.shape
    out = torch.empty((m, n), dtype=torch.int64, device=distmat.device)
    nbytes = L.lib().creid_rank_rows_workspace_bytes(m, n)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=distmat.device)
    L.check(L.lib().creid_rank_rows(L.ptr(distmat), m, n, n, L.ptr(out), L.ptr(ws), nbytes, L.stream()),
            "creid_rank_rows")
    return out


def rank_rows_eval(distmat: torch.Tensor, q_pids, g_pids, q_camids, g_camids):
    """rank_rows + the per-query half of eval_func (plain camera ids) in one pass over the distance matrix
    (creid_rank_rows_eval): returns (indices int64 [m, n], valid u8 [m], ap f64 [m], first i32 [m]) -- the index matrix is
    written for the caller but never read back by the evaluation."""
    L.require_gpu(distmat)
    assert distmat.dtype == torch.float32 and distmat.dim() == 2
    m, n = distmat.shape
    dev = distmat.device
    qp, gp, qc, gc = (_dev_i64(a, dev) for a in (q_pids, g_pids, q_camids, g_camids))
    out = torch.empty((m, n), dtype=torch.int64, device=dev)
    valid = torch.empty(m, dtype=torch.uint8, device=dev)
    ap = torch.empty(m, dtype=torch.float64, device=dev)
    first = torch.empty(m, dtype=torch.int32, device=dev)
    nbytes = L.lib().creid_rank_rows_workspace_bytes(m, n)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    L.check(L.lib().creid_rank_rows_eval(L.ptr(distmat), m, n, n, L.ptr(out), L.ptr(ws), nbytes, L.ptr(qp), L.ptr(gp), L.ptr(qc),
                                         L.ptr(gc), L.ptr(valid), L.ptr(ap), L.ptr(first), L.stream()), "creid_rank_rows_eval")
    return out, valid, ap, first


def topk_rows(distmat: torch.Tensor, k: int):
    """(indices int64 [m, k], distances fp32 [m, k]): the first k columns of np.argsort(distmat, axis=1) and the
    distances there (inference/get_similar.py:114-119), selected without sorting the whole row."""
    L.require_gpu(distmat)
    assert distmat.dtype == torch.float32 and distmat.dim() == 2
    m, n = distmat.shape
    if k > 1024 or k * 2 > n:
        idx = rank_rows(distmat)[:, :k].contiguous()
        return idx, torch.gather(distmat, 1, idx)
    dev = distmat.device
    idx = torch.empty((m, k), dtype=torch.int64, device=dev)
    dsel = torch.empty((m, k), dtype=torch.float32, device=dev)
    flags = torch.empty(m, dtype=torch.uint8, device=dev)
    L.check(L.lib().creid_topk_rows(L.ptr(distmat), m, n, n, k, L.ptr(idx), L.ptr(dsel), L.ptr(flags), L.stream()),
            "creid_topk_rows")
    bad = torch.nonzero(flags).flatten()
    if bad.numel():                                         # rows with massive ties at the k-th distance
        sub = distmat.index_select(0, bad).contiguous()
        ridx = rank_rows(sub)[:, :k].contiguous()
        idx.index_copy_(0, bad, ridx)
        dsel.index_copy_(0, bad, torch.gather(sub, 1, ridx))
    return idx, dsel


def _dev_i64(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.int64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.int64), device=device)


def _camset_masks(g_camids):
    """list of camera-id lists -> int64 bitmasks (camera ids must be < 63)."""
    out = np.zeros(len(g_camids), np.int64)
    for j, cs in enumerate(g_camids):
        for c in np.atleast_1d(cs):
            c = int(c)
            if not 0 <= c < 63:
                raise L.CreidError("camera-set evaluation supports camera ids 0..62")
            out[j] |= np.int64(1) << np.int64(c)
    return out


def eval_func_device(indices: torch.Tensor, q_pids, g_pids, q_camids, g_camids, max_rank=50, camsets=False):
    """Device half of eval_func: returns device tensors (cmc f32[max_rank], mAP f64[1], topk f64[5],
    nvalid i64[1], valid u8[m], ap f64[m], first i32[m]).  camsets: g_camids is a list of camera-id lists."""
    L.require_gpu(indices)
    dev = indices.device
    m, n = indices.shape
    if n < max_rank:  # utils/eval_reid.py:33-35
        max_rank = n
        print("Note: number of gallery samples is quite small, got {}".format(n))
    if camsets:
        g_camids = _camset_masks(g_camids)
        q_camids = np.asarray([int(np.atleast_1d(c)[0]) for c in q_camids], np.int64)
        if q_camids.size and (q_camids.min() < 0 or q_camids.max() >= 63):       # the kernel shifts a 64-bit mask by it
            raise L.CreidError("camera-set evaluation supports camera ids 0..62")
    qp, gp, qc, gc = (_dev_i64(a, dev) for a in (q_pids, g_pids, q_camids, g_camids))
    valid = torch.empty(m, dtype=torch.uint8, device=dev)
    ap = torch.empty(m, dtype=torch.float64, device=dev)
    first = torch.empty(m, dtype=torch.int32, device=dev)
    lib = L.lib()
    fn = lib.creid_cmc_ap_ranked_camsets if camsets else lib.creid_cmc_ap_ranked
    L.check(fn(L.ptr(indices), m, n, L.ptr(qp), L.ptr(gp), L.ptr(qc), L.ptr(gc), L.ptr(valid), L.ptr(ap), L.ptr(first),
               L.stream()), "creid_cmc_ap_ranked")
    cmc = torch.empty(max_rank, dtype=torch.float32, device=dev)
    mAP = torch.empty(1, dtype=torch.float64, device=dev)
    topk = torch.empty(5, dtype=torch.float64, device=dev)
    nvalid = torch.empty(1, dtype=torch.int64, device=dev)
    L.check(lib.creid_eval_reduce(L.ptr(valid), L.ptr(ap), L.ptr(first), m, max_rank, L.ptr(cmc), L.ptr(mAP),
                                  L.ptr(topk), L.ptr(nvalid), L.stream()), "creid_eval_reduce")
    return cmc, mAP, topk, nvalid, valid, ap, first


def eval_func(indices, q_pids, g_pids, q_camids, g_camids, max_rank=50, respect_camids=False):
    """utils/eval_reid.py:25-92: returns (all_cmc float32[max_rank], mAP float, all_topk float64[5],
    single_performance float64[n_valid, 3] = rows [q_idx, q_pid, AP])."""
    if not isinstance(indices, torch.Tensor):
        raise L.CreidError("eval_func needs a device tensor of ranked indices (no CPU fallback)")
    cmc, mAP, topk, nvalid, valid, ap, first = eval_func_device(indices, q_pids, g_pids, q_camids, g_camids, max_rank,
                                                                camsets=bool(respect_camids))
    valid_h = valid.cpu().numpy().astype(bool)
    vi = np.nonzero(valid_h)[0]
    qp = np.asarray(q_pids.cpu() if isinstance(q_pids, torch.Tensor) else q_pids)
    single = np.stack([vi.astype(np.float64), qp[vi].astype(np.float64), ap.cpu().numpy()[vi]], axis=1)
    return cmc.cpu().numpy(), float(mAP.item()), topk.cpu().numpy(), single


# --------------------------------------------------------------------------- streamed (metric-only) evaluation
_CAP_HINT = {}      # (queries, gallery) -> positive-list capacity of the last streamed evaluation of that shape (a speculation
                    # that R1_mAP._compute_streamed verifies; never trusted)


_LABEL_STAGE = {}   # (device, length) -> (pinned [2, length] int64 staging tensor, event of its last upload)


def _upload_labels(p, c, device):
    """[2, len] int64 device tensor of (pids, camids) through a reused page-locked staging buffer: the copy is enqueued on the
    current stream instead of blocking the host the way a pageable upload does (the evaluation's clock includes this upload;
    the device is still normalising the features while it runs).  The buffer is reused only after its previous upload has
    completed (event), whatever the caller did in between."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.from_numpy(np.stack([p, c])).to(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), len(p))
    ent = _LABEL_STAGE.get(key)
    if ent is None:
        if len(_LABEL_STAGE) > 8:
            _LABEL_STAGE.clear()
        ent = _LABEL_STAGE[key] = (torch.empty((2, len(p)), dtype=torch.int64, pin_memory=True), torch.cuda.Event())
    else:
        ent[1].synchronize()
    stage, ev = ent
    h = stage.numpy()
    np.copyto(h[0], p); np.copyto(h[1], c)
    lab = stage.to(dev, non_blocking=True)
    ev.record(torch.cuda.current_stream(dev))
    return lab


class StreamPlan:
    """Index for the streamed evaluation (csrc/stream_eval.hip): the gallery grouped by pid (CSR), every query's slot in
    it, and the per-query number of positives (same pid, different camera) which fixes the LDS list capacity `cap`.
    Queries with more than 128 positives are listed in `overflow` and must take the general (materialised) path.  Holds
    device tensors only.

    `StreamPlan.on_device(...)` (what R1_mAP uses) builds it with creid_stream_plan: one upload of the label vectors, a
    counting sort on the GPU, and an 8-byte read-back for `cap`.  The constructor is the host (numpy) construction of the
    same index -- kept for label sets whose pid range is too sparse for a dense counting sort, and as the checker of the
    device build in tests/test_stream_eval_gpu.py."""

    MAX_CAP = 128
    MAX_DENSE_RANGE = 1 << 24          # pid range the device counting sort accepts (2 x int32 scratch + int64 CSR per slot)

    @classmethod
    def on_device(cls, pids, camids, num_query, device):
        """pids / camids: host vectors [nq + ng] as R1_mAP.compute receives them (utils/reid_metric.py:112)."""
        p = np.ascontiguousarray(np.asarray(pids), dtype=np.int64)
        c = np.ascontiguousarray(np.asarray(camids), dtype=np.int64)
        nq = int(num_query)
        m, n = nq, len(p) - nq
        if n <= 0:
            return cls(p[:nq], p[nq:], c[:nq], c[nq:], device)
        pmin, pmax = int(p[nq:].min()), int(p[nq:].max())
        R = pmax - pmin + 1
        if R > cls.MAX_DENSE_RANGE:
            return cls(p[:nq], p[nq:], c[:nq], c[nq:], device)
        self = cls.__new__(cls)
        self.m, self.n = m, n
        lab = _upload_labels(p, c, device)                                        # ONE upload: [2, nq + ng] int64
        self.q_pids, self.g_pids, self.q_cams, self.g_cams = lab[0, :nq], lab[0, nq:], lab[1, :nq], lab[1, nq:]
        # one allocation for the index: [csr_off int64 (R + 1) | g_order | q_slot | n_pos | stats | scratch (2 R)] int32
        mm = max(m, 1)
        ws = torch.empty(2 * (R + 1) + n + 2 * mm + 2 + 2 * R, dtype=torch.int32, device=device)
        self.csr_off = ws[:2 * (R + 1)].view(torch.int64)
        o = 2 * (R + 1)
        self.g_order = ws[o:o + n]; o += n
        self.q_slot = ws[o:o + mm]; o += mm
        self._n_pos_dev = ws[o:o + mm]; o += mm
        self._stats = ws[o:o + 2]; o += 2
        scratch = ws[o:o + 2 * R]
        L.check(L.lib().creid_stream_plan(L.ptr(self.q_pids), L.ptr(self.g_pids), L.ptr(self.q_cams), L.ptr(self.g_cams), m, n,
                                          pmin, R, L.ptr(self.csr_off), L.ptr(self.g_order), L.ptr(self.q_slot),
                                          L.ptr(self._n_pos_dev), L.ptr(self._stats), L.ptr(scratch), L.stream()),
                "creid_stream_plan")
        self._n_pos = None
        self.cap = None                      # resolved by finish(): the only host read, 8 bytes
        return self

    def finish(self):
        """Read back {max positives, #overflow queries} (the one synchronisation of the device build) and fix `cap`."""
        if self.cap is not None:
            return self
        mx, nover = (int(v) for v in self._stats.cpu().tolist())
        cap = 2
        while cap < max(mx, 1):
            cap *= 2
        self.cap = cap
        self.overflow = np.nonzero(self.n_pos > self.MAX_CAP)[0] if nover else np.zeros(0, np.int64)
        return self

    @property
    def n_pos(self):
        if self._n_pos is None:
            self._n_pos = self._n_pos_dev[:self.m].cpu().numpy().astype(np.int64)
        return self._n_pos

    def __init__(self, q_pids, g_pids, q_camids, g_camids, device):
        qp = np.ascontiguousarray(np.asarray(q_pids), dtype=np.int64)
        gp = np.ascontiguousarray(np.asarray(g_pids), dtype=np.int64)
        qc = np.ascontiguousarray(np.asarray(q_camids), dtype=np.int64)
        gc = np.ascontiguousarray(np.asarray(g_camids), dtype=np.int64)
        self.m, self.n = len(qp), len(gp)
        order = np.argsort(gp, kind="stable").astype(np.int32)            # by pid, gallery index ascending inside
        upid, start = np.unique(gp[order], return_index=True)
        csr = np.concatenate([start, [self.n]]).astype(np.int64)
        slot = np.searchsorted(upid, qp)
        slot_c = np.minimum(slot, max(len(upid) - 1, 0))
        hit = (slot < len(upid)) & (upid[slot_c] == qp) if len(upid) else np.zeros(self.m, bool)
        same_pid = np.where(hit, csr[slot_c + 1] - csr[slot_c], 0) if len(upid) else np.zeros(self.m, np.int64)
        # same pid AND same camera (the removed entries): count through a combined key
        cmin = int(min(gc.min(initial=0), qc.min(initial=0)))
        span = int(max(gc.max(initial=0), qc.max(initial=0))) - cmin + 1
        pmin = int(min(gp.min(initial=0), qp.min(initial=0)))
        gk = (gp - pmin) * span + (gc - cmin)
        qk = (qp - pmin) * span + (qc - cmin)
        uk, cnt = np.unique(gk, return_counts=True)
        ks = np.searchsorted(uk, qk)
        ks_c = np.minimum(ks, max(len(uk) - 1, 0))
        same_cam = np.where((ks < len(uk)) & (uk[ks_c] == qk), cnt[ks_c], 0) if len(uk) else np.zeros(self.m, np.int64)
        self._n_pos = (same_pid - same_cam).astype(np.int64)
        self.overflow = np.nonzero(self._n_pos > self.MAX_CAP)[0]
        mx = int(self._n_pos[self._n_pos <= self.MAX_CAP].max(initial=1))
        cap = 2
        while cap < mx:
            cap *= 2
        self.cap = cap
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=device)
        self.q_slot = t(np.where(hit, slot_c, -1), np.int32)
        self.csr_off, self.g_order = t(csr, np.int64), t(order, np.int32)
        self.q_pids, self.g_pids, self.q_cams, self.g_cams = t(qp, np.int64), t(gp, np.int64), t(qc, np.int64), t(gc, np.int64)


def stream_eval(fq, fg, qq, gg, plan: StreamPlan):
    if plan.cap is None:
        plan.finish()
    return _stream_eval(fq, fg, qq, gg, plan)


def _stream_eval(fq, fg, qq, gg, plan: StreamPlan):
    """Per-query (valid u8[m], AP f64[m], first-match rank i32[m]) with no m x n matrix: positives' distances ->
    streamed MFMA contraction with an in-register count epilogue -> histogram prefix.  valid == 2 marks a query
    whose positive list overflowed the plan's capacity (see StreamPlan.overflow)."""
    L.require_gpu(fq, fg, qq, gg)
    assert fq.dtype == torch.float32 and fg.dtype == torch.float32
    m, n, D = fq.shape[0], fg.shape[0], fq.shape[1]
    assert (m, n) == (plan.m, plan.n)
    dev, lib, st = fq.device, L.lib(), L.stream()
    cap = plan.cap
    pos_key = torch.empty((m, cap), dtype=torch.int32, device=dev)
    pos_idx = torch.empty((m, cap), dtype=torch.int32, device=dev)
    npos = torch.empty(m, dtype=torch.int32, device=dev)
    hist = torch.zeros((m, cap), dtype=torch.int32, device=dev)
    L.check(lib.creid_stream_poslist(L.ptr(fq), L.ptr(fg), L.ptr(qq), L.ptr(gg), m, n, D, L.ptr(plan.q_slot),
                                     L.ptr(plan.csr_off), L.ptr(plan.g_order), L.ptr(plan.q_cams), L.ptr(plan.g_cams), cap,
                                     L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), st), "creid_stream_poslist")
    L.check(lib.creid_stream_count(L.ptr(fq), L.ptr(fg), L.ptr(qq), L.ptr(gg), m, n, D, L.ptr(plan.q_pids),
                                   L.ptr(plan.g_pids), cap, L.ptr(pos_key), L.ptr(pos_idx), L.ptr(npos), L.ptr(hist), st),
            "creid_stream_count")
    valid = torch.empty(m, dtype=torch.uint8, device=dev)
    ap = torch.empty(m, dtype=torch.float64, device=dev)
    first = torch.empty(m, dtype=torch.int32, device=dev)
    L.check(lib.creid_stream_finalize(L.ptr(npos), L.ptr(hist), m, cap, L.ptr(valid), L.ptr(ap), L.ptr(first), st),
            "creid_stream_finalize")
    return valid, ap, first


def eval_reduce_device(valid, ap, first, max_rank):
    """Means over valid queries (utils/eval_reid.py:86-90) on the device: (cmc f32[max_rank], mAP f64[1],
    topk f64[5], nvalid i64[1])."""
    dev = valid.device
    cmc = torch.empty(max_rank, dtype=torch.float32, device=dev)
    mAP = torch.empty(1, dtype=torch.float64, device=dev)
    topk = torch.empty(5, dtype=torch.float64, device=dev)
    nvalid = torch.empty(1, dtype=torch.int64, device=dev)
    L.check(L.lib().creid_eval_reduce(L.ptr(valid), L.ptr(ap), L.ptr(first), valid.shape[0], max_rank, L.ptr(cmc),
                                      L.ptr(mAP), L.ptr(topk), L.ptr(nvalid), L.stream()), "creid_eval_reduce")
    return cmc, mAP, topk, nvalid


class R1_mAP:
    """utils/reid_metric.py:71-151.  `pl_module` only needs `.hparams` (SOLVER.DISTANCE_FUNC,
    MODEL.USE_CENTROIDS); trainer/logger lookups of the reference are optional here."""

    def __init__(self, pl_module=None, num_query=0, max_rank=50, feat_norm=True, dist_func="euclidean",
                 compute_dtype=torch.float32, streamed=False):
        """streamed=True: metric-only evaluation that never materialises the distance / index matrices (euclidean,
        plain camera ids, fp32); `last` then holds the per-query results only.  streamed=False keeps
        `last["distmat"]` / `last["indices"]` (what the rank-index parity tests and get_similar read)."""
        self.streamed = streamed
        self.num_query = num_query
        self.max_rank = max_rank
        self.feat_norm = feat_norm
        self.pl_module = pl_module
        self.compute_dtype = compute_dtype
        if pl_module is not None and hasattr(pl_module, "hparams"):
            try:
                dist_func = pl_module.hparams.SOLVER.DISTANCE_FUNC
            except (AttributeError, KeyError):
                pass
        self.dist_name = dist_func
        self.dist_func = get_dist_func(dist_func)
        self.last = {}

    def compute(self, feats, pids, camids, respect_camids=False):
        if not isinstance(feats, torch.Tensor) or not feats.is_cuda:
            raise L.CreidError("R1_mAP.compute needs device features (no CPU fallback)")
        feats = _pad_width(feats.float().contiguous())
        nq = self.num_query
        if (self.streamed and self.dist_name == "euclidean" and not respect_camids
                and self.compute_dtype == torch.float32):
            return self._compute_streamed(feats, pids, camids)
        if self.dist_name == "euclidean":
            if self.feat_norm:
                print("The test feature is normalized")
                f, sq = l2_normalize(feats, out_dtype=self.compute_dtype, return_sqnorm=True)
            else:
                f = feats if self.compute_dtype == torch.float32 else feats.to(self.compute_dtype)
                sq = row_sqnorm(f)
            distmat = get_euclidean(f[:nq], f[nq:], sq[:nq].contiguous(), sq[nq:].contiguous())
        else:
            f = l2_normalize(feats) if self.feat_norm else feats
            distmat = self.dist_func(f[:nq].contiguous(), f[nq:].contiguous())
        pids = np.asarray(pids)
        if respect_camids:                      # (ragged list-of-lists of camera sets: keep as a list)
            indices = rank_rows(distmat)
            cmc, mAP, all_topk, single = eval_func(indices, pids[:nq], pids[nq:], camids[:nq], camids[nq:],
                                                   self.max_rank, respect_camids)
            self.last = dict(distmat=distmat, indices=indices, single_performance=single)
            return cmc, mAP, all_topk
        # np.argsort + eval_func's per-query loop in ONE pass: the ranked rows are evaluated while they are still in LDS
        camids = np.asarray(camids)
        indices, valid, ap, first = rank_rows_eval(distmat, pids[:nq], pids[nq:], camids[:nq], camids[nq:])
        max_rank = self.max_rank
        if distmat.shape[1] < max_rank:         # utils/eval_reid.py:33-35
            max_rank = distmat.shape[1]
            print("Note: number of gallery samples is quite small, got {}".format(distmat.shape[1]))
        cmc, mAP, topk, _ = eval_reduce_device(valid, ap, first, max_rank)
        pack = torch.cat([cmc.double(), mAP, topk, valid.double(), ap]).cpu().numpy()          # one read-back
        valid_h = pack[max_rank + 6:max_rank + 6 + nq] == 1
        vi = np.nonzero(valid_h)[0]
        single = np.stack([vi.astype(np.float64), pids[:nq][vi].astype(np.float64), pack[max_rank + 6 + nq:][vi]], axis=1)
        self.last = dict(distmat=distmat, indices=indices, single_performance=single, valid=valid, ap=ap, first=first)
        return pack[:max_rank].astype(np.float32), float(pack[max_rank]), pack[max_rank + 1:max_rank + 6].copy()

    def _compute_streamed(self, feats, pids, camids, plan=None, _normed=None):
        nq = self.num_query
        if _normed is not None:                 # the redo of a failed speculation: rows already normalised, nothing printed twice
            f, sq = _normed
        elif self.feat_norm:
            print("The test feature is normalized")
            f, sq = l2_normalize(feats, return_sqnorm=True)
        else:
            f, sq = feats, row_sqnorm(feats)
        pids = np.asarray(pids); camids = np.asarray(camids)
        speculative = False
        if plan is None:
            # index built on the device BEHIND the normalisation launch.  Its 8-byte read-back (finish: the positive-list
            # capacity) would be a host synchronisation in the middle of the pipeline; an evaluation of the same shape as an
            # earlier one instead ASSUMES that call's capacity, enqueues everything, and checks the assumption against the
            # plan's statistics in the final read-back (wrong -> the contraction is redone with the right capacity; a capacity
            # larger than needed gives identical results)
            plan = StreamPlan.on_device(pids, camids, nq, feats.device)
            hint = _CAP_HINT.get((plan.m, plan.n)) if os.environ.get("CREID_EVAL_SPECULATE", "1") == "1" else None
            if plan.cap is None and hint:        # 0 = "do not speculate": the last label set of this shape had overflow queries
                plan.cap, plan.overflow, speculative = hint, np.zeros(0, np.int64), True
        if plan.cap is None:
            plan.finish()
        fq, fg = f[:nq], f[nq:]
        qq, gg = sq[:nq].contiguous(), sq[nq:].contiguous()
        valid, ap, first = _stream_eval(fq, fg, qq, gg, plan)
        if len(plan.overflow):
            # queries with more positives than the LDS list holds: the general path on just those rows
            rows = torch.as_tensor(plan.overflow, device=feats.device)
            d = get_euclidean(fq.index_select(0, rows), fg, qq.index_select(0, rows), gg)
            idx = rank_rows(d)
            _, _, _, _, v2, a2, f2 = eval_func_device(idx, pids[:nq][plan.overflow], plan.g_pids, camids[:nq][plan.overflow],
                                                      plan.g_cams, self.max_rank)
            valid.index_copy_(0, rows, v2); ap.index_copy_(0, rows, a2); first.index_copy_(0, rows, f2)
        max_rank = min(self.max_rank, fg.shape[0])
        cmc, mAP, topk, _ = eval_reduce_device(valid, ap, first, max_rank)
        # ONE read-back for everything the host needs (five separate .cpu() calls are five synchronisations):
        # [cmc (max_rank) | mAP | topk (5) | valid (m) | ap (m)] as float64 (exact for the f32 / u8 members)
        stats = plan._stats.double() if speculative else torch.zeros(2, dtype=torch.float64, device=feats.device)
        pack = torch.cat([cmc.double(), mAP, topk, valid.double(), ap, stats]).cpu().numpy()
        if speculative:
            need = 2
            while need < max(int(pack[-2]), 1):
                need *= 2
            nover = int(pack[-1])
            # a label set with overflow queries (> 128 positives) cannot be speculated on (their list is only known to the
            # synchronous plan): remember that instead of a capacity, so that evaluations of this shape stop redoing
            _CAP_HINT[(plan.m, plan.n)] = 0 if nover > 0 else need
            if need > plan.cap or nover > 0:                     # the assumed capacity was too small: redo, synchronously
                plan.cap = None
                plan.finish()
                return self._compute_streamed(feats, pids, camids, plan=plan, _normed=(f, sq))
        elif getattr(plan, "_stats", None) is not None:
            _CAP_HINT[(plan.m, plan.n)] = 0 if len(plan.overflow) else plan.cap
        pack = pack[:-2]
        cmc_h = pack[:max_rank].astype(np.float32)
        mAP_h = float(pack[max_rank])
        topk_h = pack[max_rank + 1:max_rank + 6].copy()
        valid_h = pack[max_rank + 6:max_rank + 6 + nq] == 1
        ap_h = pack[max_rank + 6 + nq:]
        vi = np.nonzero(valid_h)[0]
        single = np.stack([vi.astype(np.float64), pids[:nq][vi].astype(np.float64), ap_h[vi]], axis=1)
        self.last = dict(valid=valid, ap=ap, first=first, single_performance=single, plan=plan)
        return cmc_h, mAP_h, topk_h

    def compute_chunked(self, feats, pids, camids, query_chunk=4096):
        """Galleries whose m x n matrix must not be materialised (the reference's `_commpute_batches_double` path,
        utils/reid_metric.py:93-110,126-129, chunks the gallery on the host): fp32 features go through the streamed
        kernels in ONE pass with no matrix at all; other compute dtypes process `query_chunk` query rows at a time
        (distance tile, rank, CMC/AP scan on the device, only per-query results kept)."""
        if not isinstance(feats, torch.Tensor) or not feats.is_cuda:
            raise L.CreidError("R1_mAP.compute_chunked needs device features (no CPU fallback)")
        from .parallel import merge_eval_results
        euclid = self.dist_name == "euclidean"
        feats = _pad_width(feats.float().contiguous())
        if euclid and self.compute_dtype == torch.float32:
            return self._compute_streamed(feats, pids, camids)
        nq = self.num_query
        if not euclid:
            # SOLVER.DISTANCE_FUNC = cosine (utils/reid_metric.py:51-59,93-110 works with either function): the same
            # query-chunk loop on the cosine matrix
            f = l2_normalize(feats) if self.feat_norm else feats
            sq = None
        elif self.feat_norm:
            f, sq = l2_normalize(feats, out_dtype=self.compute_dtype, return_sqnorm=True)
        else:
            f = feats if self.compute_dtype == torch.float32 else feats.to(self.compute_dtype)
            sq = row_sqnorm(f)
        g, gg = f[nq:].contiguous(), (sq[nq:].contiguous() if sq is not None else None)
        pids = np.asarray(pids); camids = np.asarray(camids)
        dev = feats.device
        gp, gc = _dev_i64(pids[nq:], dev), _dev_i64(camids[nq:], dev)
        vs, aps, firsts = [], [], []
        for lo in range(0, nq, query_chunk):
            hi = min(nq, lo + query_chunk)
            d = get_euclidean(f[lo:hi], g, sq[lo:hi].contiguous(), gg) if euclid else self.dist_func(f[lo:hi].contiguous(), g)
            idx = rank_rows(d)
            del d
            _, _, _, _, v, a, fr = eval_func_device(idx, pids[lo:hi], gp, camids[lo:hi], gc, self.max_rank)
            vs.append(v); aps.append(a); firsts.append(fr)
            del idx
        v = torch.cat(vs).cpu().numpy() > 0; a = torch.cat(aps).cpu().numpy(); fr = torch.cat(firsts).cpu().numpy()
        return merge_eval_results(v, a, fr, min(self.max_rank, g.shape[0]))
