"""Second caller of stages A + D + E: embedding extraction and similarity search with
the reference's on-disk formats.

Mirrors inference/inference_utils.py:104-159 (`_inference`, `run_inference`, `create_pid_path_index`,
`calculate_centroids`) and inference/get_similar.py:99-137 (normalise -> distance -> argsort -> top-k dict ->
`results.npy` / `query_embeddings.npy` / `query_paths.npy`), inference/create_embeddings.py:75-97
(`embeddings.npy`, `paths.npy`).  All arithmetic runs on the device through the same kernels as evaluation;
files are written with numpy exactly like the reference (dict-of-dicts pickled by np.save).
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict, List

import numpy as np
import torch

from . import _lib as L
from . import reid_metric as rm


def _inference(model, batch, use_cuda=True, normalize_with_bn=True, transform=None):
    """inference_utils.py:104-113: eval-mode backbone (+ BNNeck).  `data` is the reference's fp32 NCHW batch, or -- with
    `transform` = ReidTransforms(cfg).build_transforms(is_train=False) -- a uint8 [B, H, W, 3] batch of resized images: the test
    transform (build.py:27-31 after the Resize) then runs on the device and writes the stem convolution's operand directly."""
    model.eval()
    with torch.no_grad():
        data, _, filename = batch
        if data.dtype == torch.uint8:
            if transform is None:
                raise ValueError("uint8 image batches need `transform` (transforms.ReidTransforms(cfg).build_transforms(False))")
            if callable(transform) and getattr(transform, "_lazy_cfg", None) is not None:
                transform = transform()                          # built on the first uint8 batch only (run_inference)
            # the device-side transform IS a GPU kernel: a uint8 batch goes to the device whatever `use_cuda` says (the flag only
            # keeps the reference's meaning for float batches, which a CPU-resident model could not run here anyway)
            data = transform(data.cuda(), layout="stem", dtype=model.backbone.engine.dtype)
        else:
            data = data.cuda() if use_cuda else data
        _, global_feat = model.backbone(data)
        if normalize_with_bn:
            global_feat = model.bn(global_feat)
        return global_feat, filename


def run_inference(model, val_loader, cfg=None, print_freq=0, use_cuda=True, transform=None, macro_batch=512):
    """inference_utils.py:116-131 -> (embeddings float32 [N, D] ndarray, paths ndarray); the embeddings are
    also kept on the device in `run_inference.last_device_embeddings` for a following get_similar().  A loader of uint8
    [B, H, W, 3] batches is normalised on the device (`transform`, or the test transform built from `cfg`).

    macro_batch (round 5): consecutive loader batches are concatenated until at least this many images are waiting and embedded
    with ONE forward -- the eval-mode forward treats every image independently (BatchNorm folded to running statistics, eval-mode
    BNNeck), and every kernel variant the library may pick for another batch size produces the same bits, so the embeddings are
    IDENTICAL to the per-batch ones (tests/test_centroid_eval_gpu.py), while a forward of 512 images runs at 78 k images/s against
    65 k for the reference's TEST.IMS_PER_BATCH = 128 (profiles/r05_embed_batch_sweep.md: the persistent convolution kernels
    only pipeline across tiles when a workgroup owns more than one).  0 / None: one forward per loader batch, as the reference."""
    embs, paths = [], []
    if transform is None and cfg is not None:
        # built lazily: a float loader never touches cfg.INPUT.* (a partial cfg without those keys stays usable)
        built = {}

        def _lazy():
            if "t" not in built:
                from .transforms import ReidTransforms
                built["t"] = ReidTransforms(cfg).build_transforms(is_train=False)
            return built["t"]
        _lazy._lazy_cfg = cfg
        transform = _lazy
    pend, npend = [], 0

    def flush():
        nonlocal pend, npend
        if not pend:
            return
        data = pend[0][0] if len(pend) == 1 else torch.cat([b[0] for b in pend])
        names = [n for b in pend for n in list(b[2])]
        e, p = _inference(model, (data, None, names), use_cuda, transform=transform)
        embs.append(e.float())
        paths.extend(list(p))
        pend, npend = [], 0
    for batch in val_loader:
        data = batch[0]
        if pend and (data.dtype != pend[0][0].dtype or data.shape[1:] != pend[0][0].shape[1:] or data.device != pend[0][0].device):
            flush()                                              # batches that cannot be concatenated go separately
        pend.append(batch)
        npend += data.shape[0]
        if not macro_batch or npend >= macro_batch:
            flush()
    flush()
    dev = torch.cat(embs)
    run_inference.last_device_embeddings = dev
    return dev.cpu().numpy(), np.array(paths)


def create_pid_path_index(paths: List[str], func: Callable[[str], str]) -> Dict[str, list]:
    """inference_utils.py:134-144."""
    index: Dict[str, list] = {}
    for i, p in enumerate(paths):
        index.setdefault(func(p), []).append(i)
    return index


def calculate_centroids(embeddings, pid_path_index):
    """inference_utils.py:147-159: per-PID mean embedding -> (centroids [n_pid, D] float32, pid keys as str)."""
    keys = list(pid_path_index.keys())
    order = np.concatenate([np.asarray(pid_path_index[k], np.int64) for k in keys])
    offsets = np.concatenate([[0], np.cumsum([len(pid_path_index[k]) for k in keys])]).astype(np.int64)
    emb = embeddings if isinstance(embeddings, torch.Tensor) else torch.from_numpy(np.asarray(embeddings, np.float32))
    emb = emb.float().cuda().contiguous()
    out = torch.empty((len(keys), emb.shape[1]), dtype=torch.float32, device=emb.device)
    order_t = torch.as_tensor(order, device=emb.device)        # named: must outlive the launch
    off_t = torch.as_tensor(offsets, device=emb.device)
    L.check(L.lib().creid_gather_mean_rows(L.ptr(emb), L.ptr(order_t), L.ptr(off_t), len(keys), emb.shape[1],
                                           L.ptr(out), L.stream()), "creid_gather_mean_rows")
    return out.cpu().numpy(), np.array(keys, dtype=np.str_)


def save_embeddings(save_dir, embeddings, paths):
    """inference/create_embeddings.py:92-97."""
    d = Path(save_dir); d.mkdir(exist_ok=True, parents=True)
    np.save(d / "embeddings.npy", np.asarray(embeddings))
    np.save(d / "paths.npy", np.asarray(paths))


def load_gallery(load_dir):
    d = Path(load_dir)
    return np.load(d / "embeddings.npy", allow_pickle=True), np.load(d / "paths.npy", allow_pickle=True)


def get_similar(embeddings, paths, embeddings_gallery, paths_gallery, topk=0, normalize_features=True,
                distance_func="euclidean"):
    """inference/get_similar.py:99-125 -> {query_path: {"indices", "paths", "distances"}} (numpy arrays)."""
    q = torch.as_tensor(np.asarray(embeddings, np.float32)).cuda() if not isinstance(embeddings, torch.Tensor) else embeddings.float().cuda()
    g = torch.as_tensor(np.asarray(embeddings_gallery, np.float32)).cuda() if not isinstance(embeddings_gallery, torch.Tensor) else embeddings_gallery.float().cuda()
    if normalize_features:
        q, g = rm.l2_normalize(q.contiguous()), rm.l2_normalize(g.contiguous())
    distmat = rm.get_dist_func(distance_func)(x=q.contiguous(), y=g.contiguous()).contiguous()
    if topk:                                                   # top-k selection kernel: no full sort of the row
        indices, dist_sel = rm.topk_rows(distmat, min(int(topk), distmat.shape[1]))
        dist_sel = dist_sel.cpu().numpy()
    else:
        indices = rm.rank_rows(distmat)
        dist_sel = torch.gather(distmat, 1, indices).cpu().numpy()
    idx = indices.cpu().numpy()
    paths_gallery = np.asarray(paths_gallery)
    return {qp: {"indices": idx[i, :], "paths": paths_gallery[idx[i, :]], "distances": dist_sel[i, :]}
            for i, qp in enumerate(paths)}


def save_results(out_dir, results, embeddings, paths):
    """inference/get_similar.py:127-137."""
    d = Path(out_dir); d.mkdir(exist_ok=True, parents=True)
    np.save(d / "results.npy", results)
    np.save(d / "query_embeddings.npy", np.asarray(embeddings))
    np.save(d / "query_paths.npy", np.asarray(paths))
