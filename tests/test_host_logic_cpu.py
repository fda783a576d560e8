"""CPU: host-side logic that the GPU paths and the bench line rely on -- the numpy construction of the streamed-evaluation index
(the checker of the device build), the roofline arithmetic of the bench, the feature-width padding."""
import numpy as np
import torch


def test_host_stream_plan_against_brute_force():
    """StreamPlan (numpy construction; utils/eval_reid.py:36-65 per query): groups, positive counts, capacity and overflow
    list against a per-query brute-force count."""
    from centroids_reid_amd import reid_metric as rm
    rng = np.random.default_rng(1)
    nq, ng = 60, 900
    pids = rng.integers(0, 25, nq + ng) * 3 - 7              # non-dense, negative pids included
    cams = rng.integers(0, 4, nq + ng)
    pids[0] = 1000                                           # absent from the gallery
    pids[nq:nq + 200] = 5; pids[1] = 5                       # 200 gallery entries of one pid -> overflow for its queries
    plan = rm.StreamPlan(pids[:nq], pids[nq:], cams[:nq], cams[nq:], "cpu")
    gp, gc = pids[nq:], cams[nq:]
    n_pos = np.array([int(((gp == pids[i]) & (gc != cams[i])).sum()) for i in range(nq)])
    np.testing.assert_array_equal(plan.n_pos, n_pos)
    np.testing.assert_array_equal(plan.overflow, np.nonzero(n_pos > 128)[0])
    assert len(plan.overflow) > 0
    mx = n_pos[n_pos <= 128].max()
    assert plan.cap >= mx and plan.cap // 2 < max(mx, 2) and plan.cap & (plan.cap - 1) == 0
    slot = plan.q_slot.numpy(); csr = plan.csr_off.numpy(); order = plan.g_order.numpy()
    assert slot[0] == -1
    for i in range(1, nq):
        grp = order[csr[slot[i]]:csr[slot[i] + 1]]
        np.testing.assert_array_equal(np.sort(grp), np.nonzero(gp == pids[i])[0])
        assert (np.diff(grp) > 0).all()                      # the host build keeps gallery-index order inside a group


def test_bench_flop_counts_match_the_survey():
    """ResNet50 256x128 = 4.0533 GMAC per image forward (8.11 GFLOP), 24.32 GFLOP forward + backward; the igemm
    family of one B = 64 step (forward + data gradient of the 52 non-stem convolutions + the stem's forward) = 1.028 TFLOP;
    ResNet50(-IBN-a) 320x320 = 25.33 GFLOP per image forward."""
    from centroids_reid_amd import bench_train as bt
    fwd = bt.forward_flops(1, 256, 128)
    assert abs(fwd / 2 / 1e9 - 4.0533) < 2e-3
    assert abs(3 * fwd / 1e9 - bt.R50_FWD_BWD_GFLOP_PER_IMG) < 0.02
    assert abs(bt.forward_flops(1, 320, 320) / 1e9 - 25.33) < 0.02
    step = bt.igemm_step_flops(64, 256, 128)
    stem = 2.0 * 64 * 128 * 64 * 64 * 147
    assert abs(step - (2 * (bt.forward_flops(64, 256, 128) - stem) + stem)) < 1.0
    assert abs(step / 1e12 - 1.0278) < 1e-3
    assert len(bt.conv_shapes(64, 256, 128)) == 52


def test_feature_width_padding_changes_no_distance():
    from centroids_reid_amd import reid_metric as rm
    f = torch.randn(7, 30, generator=torch.Generator().manual_seed(0))
    p = rm._pad_width(f)
    assert p.shape == (7, 32) and torch.equal(p[:, :30], f) and float(p[:, 30:].abs().sum()) == 0.0
    assert rm._pad_width(p) is p
    d_p = (p.double()[:, None] - p.double()[None]).pow(2).sum(-1)
    d_f = (f.double()[:, None] - f.double()[None]).pow(2).sum(-1)
    assert torch.allclose(d_p, d_f, rtol=1e-13, atol=0)     # zero columns add zeros to every sum (reduction trees may differ)


def test_center_sgd_tail_adoption_is_rechecked_when_grad_is_rebound():
    """the centers' gradient rides in the tail of the Adam gradient buffer only as long as `.grad` IS a view of that
    tail.  `grad_in_adam_tail` is evaluated at every read, so a rebound gradient (zero_grad(set_to_none=True) in a foreign
    wrapper, `p.grad = ...`) sends the data-parallel sync down the explicit all-reduce instead of stepping on an unreduced
    gradient; readopt_tail() moves a rebound gradient back (values kept)."""
    import torch
    from centroids_reid_amd.solver import CenterSGD
    c = torch.nn.Parameter(torch.zeros(5, 3))
    tail = torch.zeros(16)
    opt = CenterSGD([{"params": [c], "names": ["center_loss.centers"]}], lr=0.5)
    assert opt.grad_in_adam_tail is False
    opt.adopt_tail(tail)
    assert opt.grad_in_adam_tail and c.grad.data_ptr() == tail.data_ptr()
    c.grad.fill_(2.0)
    assert float(tail[:15].sum()) == 30.0
    c.grad = torch.full((5, 3), 7.0)                    # rebound by user code
    assert opt.grad_in_adam_tail is False
    opt.readopt_tail()
    assert opt.grad_in_adam_tail and float(tail[:15].sum()) == 105.0 and float(c.grad.sum()) == 105.0
    c.grad = None                                        # zero_grad(set_to_none=True) of a foreign wrapper
    assert opt.grad_in_adam_tail is False
    opt.readopt_tail()
    assert opt.grad_in_adam_tail and float(tail.sum()) == 0.0


def test_tuned_plan_file_is_well_formed():
    """centroids-reid_amd/tuned_plans.json (what _lib registers through creid_tune_set at load time): every (kind, key) unique -- a
    duplicate would make the later entry silently win --, kinds / key lengths / plan words in the ranges the launchers decode
    (csrc/tune.hpp, conv_wgrad.hip plan_wgrad, conv_igemm.hip launch_igemm), and the merge tool additive."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "centroids-reid_amd", "tuned_plans.json")
    doc = json.load(open(path))
    plans = doc["plans"]
    keys = [(e["kind"], tuple(e["key"])) for e in plans]
    assert len(keys) == len(set(keys)) and len(plans) >= 400
    for e in plans:
        assert e["kind"] in (0, 1) and len(e["plan"]) == 3 and all(isinstance(v, int) for v in e["key"] + e["plan"])
        if e["kind"] == 0:                                   # weight gradient: (M, out_c, K, stride << 1) -> (tile rows, tile cols, splits | flags)
            assert len(e["key"]) == 4 and e["plan"][0] in (64, 128) and e["plan"][1] in (64, 128) and (e["plan"][2] & 0xffff) >= 1
        else:                                                # forward / data gradient: (M, N, K, transposed | stride << 1 | eval << 3)
            assert len(e["key"]) == 4 and e["plan"][2] in (0, 1, 2, 3, 4, 5)
            if e["plan"][2] != 5:
                assert e["plan"][0] in (64, 128) and e["plan"][1] in (2, 3, 4)
    # additive merge: a base entry is never replaced, only new keys are appended
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        base = {"plans": [{"kind": 1, "key": [1, 2, 3, 0], "plan": [64, 2, 0]}]}
        new = {"plans": [{"kind": 1, "key": [1, 2, 3, 0], "plan": [128, 3, 1]}, {"kind": 0, "key": [9, 9, 9, 2], "plan": [64, 64, 4]}]}
        for n, d in (("b", base), ("n", new)):
            json.dump(d, open(os.path.join(td, n + ".json"), "w"))
        subprocess.run([sys.executable, os.path.join(root, "tools", "merge_plans.py"), os.path.join(td, "b.json"),
                        os.path.join(td, "n.json"), os.path.join(td, "o.json")], check=True, capture_output=True)
        out = json.load(open(os.path.join(td, "o.json")))["plans"]
        assert out == [base["plans"][0], new["plans"][1]]
