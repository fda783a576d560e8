"""CPU: pin the oracle (oracle/*.py) against golden vectors produced by the imported
reference (tools/gen_golden.py).  Floating tolerances are written per assertion."""
import numpy as np
import pytest
import torch

from oracle import backbone_oracle as bo
from oracle import reid_oracle as ro

EVAL_CASES = ["eval_small", "eval_d2048", "eval_tiny_gallery"]


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_pipeline_matches_reference(golden, name):
    g = golden(name)
    nq = int(g["num_query"])
    cmc, mAP, topk, ex = ro.r1_map(torch.from_numpy(g["feats"]), g["pids"], g["camids"], nq)
    if "feats_norm" in g:
        np.testing.assert_allclose(ro.l2_normalize(torch.from_numpy(g["feats"])).numpy(), g["feats_norm"],
                                   rtol=0, atol=1e-7)
        np.testing.assert_allclose(ex["dist"].numpy(), g["distmat"], rtol=0, atol=2e-6)
    # gap-designed fixture: rank indices must be BIT-EXACT
    np.testing.assert_array_equal(ex["indices"], g["indices"])
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)
    assert abs(mAP - float(g["mAP"])) < 1e-12
    np.testing.assert_allclose(topk, g["topk"], rtol=0, atol=1e-12)
    single = g["single"]                      # rows [q_idx, q_pid, AP] of valid queries
    valid_idx = np.nonzero(ex["valid"])[0]
    np.testing.assert_array_equal(valid_idx, single[:, 0].astype(np.int64))
    np.testing.assert_allclose(ex["ap"][valid_idx], single[:, 2], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", EVAL_CASES)
def test_eval_market_on_reference_indices(golden, name):
    """Integer stage E alone, fed the reference's own argsort output."""
    g = golden(name)
    nq = int(g["num_query"])
    cmc, mAP, topk, _ = ro.eval_market(g["indices"], g["pids"][:nq], g["pids"][nq:],
                                       g["camids"][:nq], g["camids"][nq:])
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)
    assert abs(mAP - float(g["mAP"])) < 1e-12
    np.testing.assert_allclose(topk, g["topk"], rtol=0, atol=1e-12)


def test_val_centroids(golden):
    g = golden("eval_centroids")
    nq = int(g["num_query"])
    emb, labels, cams = ro.val_centroids(torch.from_numpy(g["feats"]), g["pids"], g["camids"], nq)
    np.testing.assert_allclose(emb.numpy(), g["cent_emb"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(labels, g["cent_labels"])
    np.testing.assert_array_equal(cams, g["cent_camids"])
    cmc, mAP, topk, _ = ro.r1_map(emb, labels, cams, nq)
    assert abs(mAP - float(g["mAP"])) < 1e-9
    np.testing.assert_allclose(cmc, g["cmc"], atol=1e-7)


@pytest.mark.parametrize("name", ["losses_n64_d128", "losses_n32_d2048"])
def test_losses(golden, name):
    g = golden(name)
    x = torch.from_numpy(g["x"]); labels = torch.from_numpy(g["labels"])
    d = ro.euclidean_dist(x, x)
    # the diagonal is sqrt(clamp(rounding noise)) -- implementation-defined, never selected by mining
    off = ~np.eye(len(x), dtype=bool)
    np.testing.assert_allclose(d.numpy()[off], g["dist"][off], rtol=1e-5, atol=1e-5)
    ap, an, pi, ni = ro.hard_example_mining(d, labels)
    np.testing.assert_array_equal(pi.numpy(), g["p_inds"]); np.testing.assert_array_equal(ni.numpy(), g["n_inds"])
    for tag, margin, m in (("m05", 0.5, None), ("soft", None, None), ("m05_mask", 0.5, g["mask"])):
        xt = x.clone().requires_grad_(True)
        loss, ap, an = ro.triplet_loss(xt, labels, margin, None if m is None else torch.from_numpy(m))
        loss.backward()
        assert abs(loss.item() - float(g[f"trip_{tag}_loss"])) < 1e-5
        np.testing.assert_allclose(ap.detach().numpy(), g[f"trip_{tag}_ap"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xt.grad.numpy(), g[f"trip_{tag}_grad"], rtol=1e-4, atol=1e-6)
    xt = x.clone().requires_grad_(True); c = torch.from_numpy(g["centers"]).requires_grad_(True)
    l = ro.center_loss(xt, labels, c); l.backward()
    assert abs(l.item() - float(g["center_loss"])) < 1e-5 * abs(float(g["center_loss"]))
    np.testing.assert_allclose(xt.grad.numpy(), g["center_grad_x"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c.grad.numpy(), g["center_grad_c"], rtol=1e-4, atol=1e-6)
    lt = torch.from_numpy(g["logits"]).requires_grad_(True)
    l = ro.xent_label_smooth(lt, labels); l.backward()
    assert abs(l.item() - float(g["xent_loss"])) < 1e-5
    np.testing.assert_allclose(lt.grad.numpy(), g["xent_grad"], rtol=1e-4, atol=1e-7)


HEADS = ["heads_p16k4_d128", "heads_p16k4_d128_fake1", "heads_p16k4_d128_fake2", "heads_p8k4_d2048"]


@pytest.mark.parametrize("name", HEADS)
def test_heads_step(golden, name):
    g = golden(name)
    P, K = int(g["P"]), int(g["K"])
    feats = torch.from_numpy(g["feats"]).requires_grad_(True)
    labels = torch.from_numpy(g["labels"]); is_real = torch.from_numpy(g["is_real"])
    centers = torch.from_numpy(g["centers0"]).requires_grad_(True)
    fc = torch.from_numpy(g["fc0"]).requires_grad_(True)
    bn_w = torch.from_numpy(g["bn_w0"]).requires_grad_(True)
    D = feats.shape[1]
    rm, rv = torch.zeros(D), torch.ones(D)
    masks, _ = ro.create_masks_train(g["labels"])
    np.testing.assert_array_equal(masks, g["masks"])
    out = ro.ctl_heads(feats, labels, is_real, bn_w, torch.from_numpy(g["bn_b0"]), rm, rv, fc, centers,
                       P, K, margin=float(g["margin"]))
    out["total"].backward()
    for n in ("query_xent", "query_triplet", "query_center", "centroid_triplet"):
        assert abs(out[n].item() - float(g[f"s0_{n}"])) < 2e-5, n
    assert abs(out["total"].item() - float(g["s0_loss_total"])) < 2e-5
    for n in ("step_dist_ap", "step_dist_an", "l2_mean_centroid"):
        assert abs(out[n].item() - float(g[f"s0_{n}"])) < 2e-5, n
    np.testing.assert_allclose(feats.grad.numpy(), g["s0_grad_features"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(fc.grad.numpy(), g["s0_grad_fc"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(bn_w.grad.numpy(), g["s0_grad_bn_w"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(centers.grad.numpy() / 5e-4, g["s0_grad_centers_scaled"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ro.center_sgd_step(centers.detach(), centers.grad).numpy(), g["s0_centers_after"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rm.numpy(), g["s0_bn_rm_after"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.numpy(), g["s0_bn_rv_after"], rtol=1e-5, atol=1e-7)
    # Adam (lr warm-up: epoch 0 -> 0.1 * BASE_LR, train_ctl_model.py:41-49)
    lr = 0.1 * float(g["base_lr"])
    p1, _, _ = ro.adam_step(fc.detach(), fc.grad, torch.zeros_like(fc), torch.zeros_like(fc), 1, lr)
    # (atol: Adam's first step is lr * g / (|g| + eps); for the few classifier entries whose gradient is ~eps the quotient moves by
    #  1e-2 of a step with the last bit of g, and torch-CPU's fp32 reductions differ in that bit between host CPUs -- seen on the GPU
    #  box's host: 1 of 7680 entries off by 2.5e-7 = 0.7 % of one 3.5e-5 step)
    np.testing.assert_allclose(p1.numpy(), g["s0_fc_after"], rtol=1e-5, atol=5e-7)


@pytest.mark.parametrize("name", ["backbone_r50_2x256x128", "backbone_r50ibn_2x64x64", "backbone_r101_2x64x64",
                                  "backbone_r152_2x64x64", "backbone_r101ibn_2x64x64"])
def test_backbone_oracle(golden, name):
    g = golden(name)
    arch = str(g["arch"]); B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    torch.set_num_threads(8)
    sd = bo.make_state_dict(arch, 1, seed=1234)
    x = bo.synthetic_images(B, H, W, seed=7)
    with torch.no_grad():
        y, feat = bo.backbone_forward(x, sd, arch, 1, training=False)
    np.testing.assert_allclose(feat.numpy(), g["eval_feat"], rtol=1e-4, atol=1e-5)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point
              and not k.endswith(("running_mean", "running_var"))}
    sd2 = {**{k: v.clone() for k, v in sd.items()}, **params}
    y, feat = bo.backbone_forward(x, sd2, arch, 1, training=True)
    np.testing.assert_allclose(feat.detach().numpy(), g["train_feat"], rtol=1e-4, atol=1e-5)
    coef = torch.from_numpy(np.random.default_rng(99).standard_normal((B, 2048)).astype(np.float32))
    (feat * coef).sum().backward()
    np.testing.assert_allclose(sd2["bn1.running_mean"].numpy(), g["bn1_rm"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd2["layer4.2.bn3.running_var"].numpy(), g["l4_bn3_rv"], rtol=1e-4, atol=1e-6)
    gc1 = params["conv1.weight"].grad.numpy()
    scale = np.abs(g["grad_conv1"]).max()
    np.testing.assert_allclose(gc1, g["grad_conv1"], rtol=2e-3, atol=2e-3 * scale)
    gl4 = params["layer4.2.conv3.weight"].grad[:16, :, 0, 0].numpy()
    np.testing.assert_allclose(gl4, g["grad_l4_conv3_slice"], rtol=2e-3, atol=2e-3 * np.abs(gl4).max())


def test_val_centroids_camera_sets(golden):
    """respect_camids=True (modelling/bases.py:205-253 + utils/eval_reid.py:51-55) against the reference."""
    g = golden("eval_camsets")
    nq = int(g["num_query"])
    emb, labels, camsets = ro.val_centroids_camera(torch.from_numpy(g["feats"]), g["pids"], g["camids"], nq)
    np.testing.assert_allclose(emb.numpy(), g["cent_emb"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(labels, g["cent_labels"])
    ref_sets = [[int(c) for c in row if c >= 0] for row in g["cent_camsets"]]
    assert camsets[nq:] == ref_sets
    cmc, mAP, topk, ex = ro.eval_market_camsets(g["indices"], labels[:nq], labels[nq:], g["camids"][:nq], camsets[nq:])
    assert abs(mAP - float(g["mAP"])) < 1e-12
    np.testing.assert_allclose(cmc, g["cmc"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(topk, g["topk"], rtol=0, atol=1e-12)
    # and the oracle's own ranking of the centroid gallery reproduces the reference's indices
    f = ro.l2_normalize(emb)
    idx = ro.rank_rows(ro.sqdist_matrix(f[:nq], f[nq:]))
    gap_ok = np.ones_like(idx, bool)
    ds = np.take_along_axis(ro.sqdist_matrix(f[:nq], f[nq:]).numpy(), idx, 1)
    gap = np.diff(ds, axis=1) > 1e-6
    gap_ok[:, 1:] &= gap; gap_ok[:, :-1] &= gap
    np.testing.assert_array_equal(idx[gap_ok], g["indices"][gap_ok])


def test_inference_golden_vs_oracle(golden):
    """inference/inference_utils.py:134-159 + inference/get_similar.py:97-125 recorded from the reference: the oracle's
    normalise -> squared-L2 -> stable rank pipeline reproduces the top-k indices and distances."""
    g = golden("inference")
    nq, topk = int(g["num_query"]), int(g["topk"])
    f = torch.from_numpy(g["feats"])
    q, gal = ro.l2_normalize(f[:nq]), ro.l2_normalize(f[nq:])
    d = ro.sqdist_matrix(q, gal)
    idx = ro.rank_rows(d)[:, :topk]
    np.testing.assert_array_equal(idx, g["indices"])
    np.testing.assert_allclose(np.take_along_axis(d.numpy(), idx, 1), g["distances"], rtol=0, atol=2e-6)
    # per-pid index + centroids (plain numpy restatement of the two helper functions)
    keys = [p.split("/")[-1].split("_")[0] for p in g["gallery_paths"]]
    order = list(dict.fromkeys(keys))
    assert order == list(g["index_keys"]) == list(g["centroid_keys"])
    off = np.concatenate([[0], np.cumsum(g["index_sizes"])])
    for i, k in enumerate(order):
        members = np.nonzero(np.asarray(keys) == k)[0]
        np.testing.assert_array_equal(members, g["index_flat"][off[i]:off[i + 1]])
        np.testing.assert_allclose(f[nq:].numpy()[members].sum(0) / len(members), g["centroids"][i], rtol=0, atol=1e-6)


# ---- round 2: boundary surface (cosine / normalize_feature / general distances / ragged masks)
def _mask_sets(g):
    for i in range(int(g["n_mask_sets"])):
        lens = g[f"mask{i}_lists_len"]
        flat = g[f"mask{i}_lists_flat"]
        lists, o = [], 0
        for n in lens:
            lists.append(flat[o:o + n].tolist()); o += n
        yield g[f"mask{i}_labels"], g[f"mask{i}_masks"], lists


def test_create_masks_ragged_oracle_and_product(golden):
    """create_masks_train is host logic: the oracle AND the product's own rewrite are checked against the
    reference's output on ragged / interleaved label vectors (incl. the cumsum[-1] quirk for a short first PID)."""
    from centroids_reid_amd.bases import ModelBase
    g = golden("surface_r2")
    for labels, masks, lists in _mask_sets(g):
        mo, lo = ro.create_masks_train(labels)
        np.testing.assert_array_equal(mo, masks)
        assert [list(map(int, v)) for v in lo] == lists
        mp, lp = ModelBase.create_masks_train(torch.from_numpy(labels))
        assert mp.dtype == torch.bool
        np.testing.assert_array_equal(mp.numpy(), masks)
        assert lp == lists


def test_triplet_cosine_normalize_oracle(golden):
    g = golden("surface_r2")
    x = torch.from_numpy(g["x"]); labels = torch.from_numpy(g["labels"])
    for tag, margin, dist, norm, m in (("cos_m05", 0.5, "cosine", False, None), ("cos_soft", None, "cosine", False, None),
                                       ("cos_m05_mask", 0.5, "cosine", False, g["mask"]),
                                       ("euc_norm", 0.5, "euclidean", True, None), ("cos_norm", 0.3, "cosine", True, None)):
        xt = x.clone().requires_grad_(True)
        loss, ap, an = ro.triplet_loss(xt, labels, margin, None if m is None else torch.from_numpy(m), dist, norm)
        loss.backward()
        assert abs(loss.item() - float(g[f"trip_{tag}_loss"])) < 1e-6
        np.testing.assert_allclose(ap.detach().numpy(), g[f"trip_{tag}_ap"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(an.detach().numpy(), g[f"trip_{tag}_an"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(xt.grad.numpy(), g[f"trip_{tag}_grad"], rtol=1e-4, atol=1e-7)
    y = torch.from_numpy(g["y"])
    np.testing.assert_allclose(ro.euclidean_dist(x, y).numpy(), g["xy_euc_dist"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ro.cosine_dist(x, y).numpy(), g["xy_cos_dist"], rtol=1e-5, atol=1e-6)
    ap, an, pi, ni = ro.hard_example_mining(torch.from_numpy(g["mine_dist"]), labels)
    np.testing.assert_array_equal(pi.numpy(), g["mine_pi"]); np.testing.assert_array_equal(ni.numpy(), g["mine_ni"])
    np.testing.assert_array_equal(ap.numpy(), g["mine_ap"]); np.testing.assert_array_equal(an.numpy(), g["mine_an"])


@pytest.mark.parametrize("name", ["backbone_r18_2x128x64", "backbone_r34_2x128x64"])
def test_basic_block_backbone_oracle(golden, name):
    """resnet18 / resnet34 (modelling/baseline.py:56-65, modelling/backbones/resnet.py:22-48 BasicBlock): the oracle against
    recordings of the reference's own ResNet(block=BasicBlock) on the same seeded weights."""
    g = golden(name)
    arch = str(g["arch"]); B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    torch.set_num_threads(8)
    sd = bo.make_state_dict(arch, 1, seed=1234)
    assert len(sd) == int(g["n_keys"])
    x = bo.synthetic_images(B, H, W, seed=7)
    with torch.no_grad():
        y, feat = bo.backbone_forward(x, sd, arch, 1, training=False)
    assert feat.shape == (B, 512)
    np.testing.assert_allclose(feat.numpy(), g["eval_feat"], rtol=1e-4, atol=1e-5)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point
              and not k.endswith(("running_mean", "running_var"))}
    sd2 = {**{k: v.clone() for k, v in sd.items()}, **params}
    y, feat = bo.backbone_forward(x, sd2, arch, 1, training=True)
    np.testing.assert_allclose(feat.detach().numpy(), g["train_feat"], rtol=1e-4, atol=1e-5)
    coef = torch.from_numpy(np.random.default_rng(99).standard_normal((B, 512)).astype(np.float32))
    (feat * coef).sum().backward()
    last = "layer4.%d.bn2" % (bo.ARCH_LAYERS[arch][3] - 1)
    np.testing.assert_allclose(sd2[last + ".running_var"].numpy(), g["l4_bn2_rv"], rtol=1e-4, atol=1e-6)
    gc1 = params["conv1.weight"].grad.numpy()
    np.testing.assert_allclose(gc1, g["grad_conv1"], rtol=2e-3, atol=2e-3 * np.abs(g["grad_conv1"]).max())
    gds = params["layer2.0.downsample.0.weight"].grad[:, :, 0, 0].numpy()
    np.testing.assert_allclose(gds, g["grad_l2_ds"], rtol=2e-3, atol=2e-3 * np.abs(gds).max())
