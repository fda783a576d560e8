"""GPU parity: centroid-based validation (modelling/bases.py:179-297) in both modes and the inference /
similarity-search caller (inference/inference_utils.py, inference/get_similar.py) through the C ABI."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _holder(nq, keep_camid=False):
    from centroids_reid_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.MODEL.USE_CENTROIDS = True
    cfg.MODEL.KEEP_CAMID_CENTROIDS = keep_camid
    cfg.num_query = nq
    return SimpleNamespace(hparams=cfg, trainer=SimpleNamespace(logger=None, current_epoch=0))


def test_val_centroids_plain_golden(golden):
    from centroids_reid_amd.bases import ModelBase
    from centroids_reid_amd.reid_metric import R1_mAP
    g = golden("eval_centroids")
    nq = int(g["num_query"])
    emb, labels, cams = ModelBase.validation_create_centroids(_holder(nq), torch.from_numpy(g["feats"]).cuda(),
                                                              g["pids"], g["camids"])
    np.testing.assert_allclose(emb.cpu().numpy(), g["cent_emb"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(labels, g["cent_labels"])
    np.testing.assert_array_equal(cams, g["cent_camids"])
    cmc, mAP, topk = R1_mAP(num_query=nq).compute(emb, labels, cams)
    assert abs(mAP - float(g["mAP"])) < 1e-9
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)


def test_val_centroids_camera_sets_golden(golden):
    from centroids_reid_amd.bases import ModelBase
    from centroids_reid_amd.reid_metric import R1_mAP
    g = golden("eval_camsets")
    nq = int(g["num_query"])
    emb, labels, camsets = ModelBase.validation_create_centroids(
        _holder(nq, True), torch.from_numpy(g["feats"]).cuda(), g["pids"], g["camids"], respect_camids=True)
    np.testing.assert_allclose(emb.cpu().numpy(), g["cent_emb"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(labels, g["cent_labels"])
    assert [list(map(int, s)) for s in camsets[nq:]] == [[int(c) for c in r if c >= 0] for r in g["cent_camsets"]]
    metric = R1_mAP(num_query=nq)
    cmc, mAP, topk = metric.compute(emb, labels, camsets, respect_camids=True)
    np.testing.assert_array_equal(metric.last["indices"].cpu().numpy(), g["indices"])
    assert abs(mAP - float(g["mAP"])) < 1e-12
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(topk, g["topk"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(metric.last["single_performance"][:, 2], g["single"][:, 2], rtol=0, atol=1e-12)


def test_camset_scan_integer_exact_vs_oracle():
    from centroids_reid_amd import reid_metric as rm
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(31)
    nq, ng = 130, 1999
    idx = np.stack([rng.permutation(ng) for _ in range(nq)]).astype(np.int64)
    qp = rng.integers(0, 40, nq); gp = rng.integers(0, 38, ng)
    qc = rng.integers(0, 8, nq)
    gsets = [sorted(set(rng.integers(0, 8, rng.integers(1, 5)).tolist())) for _ in range(ng)]
    cmc, mAP, topk, single = rm.eval_func(torch.from_numpy(idx).cuda(), qp, gp, [[int(c)] for c in qc], gsets,
                                          respect_camids=True)
    cmc_o, mAP_o, topk_o, ex = ro.eval_market_camsets(idx, qp, gp, qc, gsets)
    assert (~ex["valid"]).sum() > 0
    np.testing.assert_array_equal(single[:, 0].astype(np.int64), np.nonzero(ex["valid"])[0])
    np.testing.assert_allclose(single[:, 2], ex["ap"][ex["valid"]], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(cmc, cmc_o)
    assert abs(mAP - mAP_o) < 1e-13


def test_get_val_metrics_routes_camera_sets(golden, capsys):
    from centroids_reid_amd.bases import ModelBase
    g = golden("eval_camsets")
    nq = int(g["num_query"])
    h = _holder(nq, True)
    emb, labels, camsets = ModelBase.validation_create_centroids(h, torch.from_numpy(g["feats"]).cuda(), g["pids"],
                                                                 g["camids"], respect_camids=True)
    out = ModelBase.get_val_metrics(h, emb, labels, camsets)
    assert abs(out["mAP"] - float(g["mAP"])) < 1e-12
    assert abs(out["Top-1"] - float(g["topk"][0])) < 1e-12


# ------------------------------------------------------------------ inference / similarity search
def test_get_similar_matches_oracle(tmp_path):
    from centroids_reid_amd import inference as inf
    from oracle import reid_oracle as ro
    rng = np.random.default_rng(41)
    q = rng.standard_normal((37, 256)).astype(np.float32)
    gal = rng.standard_normal((900, 256)).astype(np.float32)
    qpaths = np.array([f"q/{i:04d}.jpg" for i in range(37)])
    gpaths = np.array([f"g/{i % 90:03d}_{i:05d}.jpg" for i in range(900)])
    res = inf.get_similar(q, qpaths, gal, gpaths, topk=20)
    d = ro.sqdist_matrix(ro.l2_normalize(torch.from_numpy(q)), ro.l2_normalize(torch.from_numpy(gal))).numpy()
    order = np.argsort(d, axis=1, kind="stable")
    assert list(res.keys()) == list(qpaths)
    for i, p in enumerate(qpaths):
        r = res[p]
        assert r["indices"].shape == (20,) and r["paths"].shape == (20,) and r["distances"].shape == (20,)
        ds = d[i, order[i]]
        safe = np.ones(20, bool)
        gap = np.diff(ds[:21]) > 4e-6
        safe &= gap[:20]; safe[1:] &= gap[:19]
        np.testing.assert_array_equal(r["indices"][safe], order[i, :20][safe])
        np.testing.assert_array_equal(r["paths"], gpaths[r["indices"]])
        np.testing.assert_allclose(r["distances"], d[i, r["indices"]], rtol=0, atol=3e-6)
        assert (np.diff(r["distances"]) >= 0).all()
    # on-disk formats of inference/get_similar.py:127-137 and create_embeddings.py:92-97 round-trip
    inf.save_results(tmp_path / "out", res, q, qpaths)
    back = np.load(tmp_path / "out" / "results.npy", allow_pickle=True).item()
    np.testing.assert_array_equal(back[qpaths[3]]["indices"], res[qpaths[3]]["indices"])
    np.testing.assert_array_equal(np.load(tmp_path / "out" / "query_paths.npy"), qpaths)
    inf.save_embeddings(tmp_path / "gal", gal, gpaths)
    e2, p2 = inf.load_gallery(tmp_path / "gal")
    np.testing.assert_array_equal(e2, gal); np.testing.assert_array_equal(p2, gpaths)
    # cosine distance and un-normalised / full-length variants
    res_c = inf.get_similar(q, qpaths, gal, gpaths, topk=0, distance_func="cosine")
    dc = ro.cosine_dist_matrix(ro.l2_normalize(torch.from_numpy(q)), ro.l2_normalize(torch.from_numpy(gal))).numpy()
    assert res_c[qpaths[0]]["indices"].shape == (900,)
    np.testing.assert_allclose(res_c[qpaths[5]]["distances"], dc[5, res_c[qpaths[5]]["indices"]], rtol=0, atol=3e-6)


def test_calculate_centroids_matches_numpy():
    from centroids_reid_amd import inference as inf
    rng = np.random.default_rng(43)
    emb = rng.standard_normal((500, 128)).astype(np.float32)
    paths = [f"gallery/{rng.integers(0, 40):03d}_c{rng.integers(1, 7)}_{i:05d}.jpg" for i in range(500)]
    index = inf.create_pid_path_index(paths, lambda p: p.split("/")[-1].split("_")[0])
    assert sum(len(v) for v in index.values()) == 500 and list(index.keys())[0] == paths[0].split("/")[-1][:3]
    cents, keys = inf.calculate_centroids(emb, index)
    assert cents.dtype == np.float32 and keys.dtype.kind == "U" and list(keys) == list(index.keys())
    for k, c in zip(keys, cents):
        np.testing.assert_allclose(c, emb[index[k]].astype(np.float64).mean(0), rtol=0, atol=2e-6)


def test_run_inference_uses_bnneck():
    from centroids_reid_amd import inference as inf

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = lambda x: (None, x.flatten(1)[:, :16] * 2.0)
            self.bn = lambda f: f + 1.0

    model = Stub()
    x = torch.arange(4 * 3 * 4 * 4, dtype=torch.float32).reshape(4, 3, 4, 4)
    loader = [(x[:2], None, ["a", "b"]), (x[2:], None, ["c", "d"])]
    emb, paths = inf.run_inference(model, loader)
    np.testing.assert_array_equal(emb, (x.flatten(1)[:, :16] * 2 + 1).numpy())
    assert list(paths) == ["a", "b", "c", "d"]


def test_inference_golden(golden):
    """Against the reference's own inference helpers (tests/golden/inference.npz, tools/gen_golden.py inference)."""
    from centroids_reid_amd import inference as inf
    g = golden("inference")
    nq, topk = int(g["num_query"]), int(g["topk"])
    f = g["feats"]
    res = inf.get_similar(f[:nq], g["query_paths"], f[nq:], g["gallery_paths"], topk=topk)
    assert list(res.keys()) == list(g["query_paths"])
    for i, p in enumerate(g["query_paths"]):
        np.testing.assert_array_equal(res[p]["indices"], g["indices"][i])             # gap-designed gallery: bit-exact
        np.testing.assert_array_equal(res[p]["paths"], g["gallery_paths"][g["indices"][i]])
        np.testing.assert_allclose(res[p]["distances"], g["distances"][i], rtol=0, atol=3e-6)
    index = inf.create_pid_path_index(list(g["gallery_paths"]), lambda p: p.split("/")[-1].split("_")[0])
    assert list(index.keys()) == list(g["index_keys"])
    np.testing.assert_array_equal(np.concatenate([np.asarray(v) for v in index.values()]), g["index_flat"])
    cents, keys = inf.calculate_centroids(f[nq:], index)
    assert list(keys) == list(g["centroid_keys"]) and keys.dtype == g["centroid_keys"].dtype
    np.testing.assert_allclose(cents, g["centroids"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_run_inference_macro_batches_are_bit_identical_to_per_batch(dtype):
    """inference.run_inference(macro_batch=...) (round 5): consecutive loader batches concatenated into one eval-mode forward give
    EXACTLY the embeddings of one forward per batch -- every image is independent in eval mode and every kernel variant another
    batch size may select produces the same bits -- for a real CTLModel, ragged last batch included, uint8 loaders too."""
    from centroids_reid_amd import inference as inf
    from centroids_reid_amd.bench_train import make_model
    torch.manual_seed(3)
    model = make_model(num_classes=16, dtype=dtype).eval()
    gen = torch.Generator().manual_seed(5)
    sizes = [40, 40, 40, 23]
    loader = [(torch.randn((n, 3, 128, 64), generator=gen), None, [f"img_{i}_{j}.jpg" for j in range(n)]) for i, n in enumerate(sizes)]
    e0, p0 = inf.run_inference(model, loader, macro_batch=0)
    assert e0.shape == (sum(sizes), 2048)
    for mb in (64, 100, 512):
        e1, p1 = inf.run_inference(model, loader, macro_batch=mb)
        assert list(p1) == list(p0)
        np.testing.assert_array_equal(e1, e0)
    assert np.isfinite(e0).all() and np.abs(e0).max() > 0
