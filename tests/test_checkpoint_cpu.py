"""CPU: checkpoint layout contract: our CTLModel's state_dict has exactly the
reference's keys and shapes (R50: 325, IBN-a: 353), and the PL-style checkpoint dict round-trips."""
import numpy as np
import pytest
import torch


def _cfg(arch):
    from centroids_reid_amd.config import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.MODEL.PRETRAINED = False
    cfg.MODEL.NAME = arch
    return cfg


@pytest.mark.parametrize("arch,n", [("resnet50", 325), ("resnet50_ibn_a", 353), ("resnet101", 631), ("resnet152", 937),
                                    ("resnet101_ibn_a", 693)])
def test_state_dict_layout_matches_reference(golden, arch, n):
    from centroids_reid_amd.train_ctl_model import CTLModel
    g = golden("ckpt_keys")
    m = CTLModel(_cfg(arch), num_classes=751, num_query=10)
    sd = m.state_dict()
    assert len(sd) == n == len(g[f"{arch}_keys"])
    assert list(sd.keys()) == [str(k) for k in g[f"{arch}_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g[f"{arch}_shapes"]]


def test_checkpoint_roundtrip(tmp_path):
    from centroids_reid_amd.train_ctl_model import CTLModel
    m = CTLModel(_cfg("resnet50"), num_classes=37, num_query=5)
    ck = m.checkpoint_dict(epoch=3, global_step=120)
    assert {"state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "epoch", "global_step", "callbacks"} <= set(ck)
    assert ck["hyper_parameters"]["num_classes"] == 37 and ck["hyper_parameters"]["SOLVER"]["MARGIN"] == 0.5
    p = tmp_path / "epoch=3.ckpt"
    m.save_checkpoint(p, 3, 120)
    m2 = CTLModel.load_from_checkpoint(p)
    assert m2.hparams.num_classes == 37 and m2.hparams.MODEL.NAME == "resnet50"
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # a reference-style pretrained backbone file ('base.' / 'backbone.base.' prefixes) loads through load_param
    sd = {"state_dict": {"backbone.base." + k: v for k, v in m.backbone.base.state_dict().items()}}
    torch.save(sd, tmp_path / "pre.pth")
    m3 = CTLModel(_cfg("resnet50"), num_classes=37, num_query=5)
    m3.backbone.base.load_param(str(tmp_path / "pre.pth"))
    assert torch.equal(m3.backbone.base.layer3[2].conv2.weight, m.backbone.base.layer3[2].conv2.weight)


def test_optimizer_step_hook_warms_up_lr():
    """ModelBase.optimizer_step (modelling/bases.py:102-133): linear warm-up of the stepped optimiser's lr, then the
    step (closure first).  Host logic only: a recording stand-in optimiser."""
    from centroids_reid_amd.config import get_cfg_defaults
    from centroids_reid_amd.train_ctl_model import CTLModel
    cfg = get_cfg_defaults()
    cfg.MODEL.PRETRAINED = False
    model = CTLModel(cfg, num_classes=5, num_query=0)

    class Rec:
        def __init__(self):
            self.param_groups = [{"lr": 123.0}, {"lr": 456.0}]
            self.calls = []

        def step(self, closure=None):
            self.calls.append("step")

    hp = model.hparams
    assert hp.SOLVER.USE_WARMUP_LR and hp.SOLVER.WARMUP_EPOCHS == 10
    o = Rec()
    model.optimizer_step(epoch=3, batch_idx=0, optimizer=o, optimizer_idx=0, optimizer_closure=lambda: o.calls.append("closure"))
    assert o.calls == ["closure", "step"]
    assert all(abs(pg["lr"] - 0.4 * hp.SOLVER.BASE_LR) < 1e-12 for pg in o.param_groups)
    o = Rec()
    model.optimizer_step(epoch=10, batch_idx=0, optimizer=o, optimizer_idx=0)
    assert o.calls == ["step"] and o.param_groups[0]["lr"] == 123.0          # past the warm-up: untouched


def test_load_checkpoint_with_pickled_pl_and_yacs_classes(tmp_path):
    """A checkpoint as the reference writes it: hyper_parameters pickled as pytorch-lightning AttributeDict holding
    yacs CfgNode sub-trees.  Neither package is installed here; load_from_checkpoint must still open it."""
    import importlib
    import sys
    import types
    import torch
    from centroids_reid_amd.config import get_cfg_defaults
    from centroids_reid_amd.train_ctl_model import CTLModel
    for m in ("pytorch_lightning", "yacs"):
        try:
            importlib.import_module(m)
            import pytest
            pytest.skip(f"{m} is installed: the stand-in path is not exercised")
        except ImportError:
            pass
    cfg = get_cfg_defaults()
    cfg.MODEL.PRETRAINED = False
    model = CTLModel(cfg, num_classes=7, num_query=3)
    ck = model.checkpoint_dict(epoch=4, global_step=99)
    # re-wrap the hyper parameters in classes that live under the reference's module paths (as PL / yacs pickle them)
    mods = {}
    for name in ("pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing", "yacs", "yacs.config"):
        mods[name] = types.ModuleType(name)
    AD = type("AttributeDict", (dict,), {"__module__": "pytorch_lightning.utilities.parsing"})
    CN = type("CfgNode", (dict,), {"__module__": "yacs.config"})
    mods["pytorch_lightning.utilities.parsing"].AttributeDict = AD
    mods["yacs.config"].CfgNode = CN
    hp = AD({k: (CN(v) if isinstance(v, dict) else v) for k, v in ck["hyper_parameters"].items()})
    ck["hyper_parameters"] = hp
    path = tmp_path / "ref_style.ckpt"
    sys.modules.update(mods)
    try:
        torch.save(ck, path)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    assert "yacs" not in sys.modules and "pytorch_lightning" not in sys.modules
    m2 = CTLModel.load_from_checkpoint(str(path))
    assert "yacs" not in sys.modules and "pytorch_lightning" not in sys.modules          # stand-ins removed again
    assert m2.hparams.num_classes == 7 and m2.hparams.num_query == 3 and m2.hparams.SOLVER.MARGIN == cfg.SOLVER.MARGIN
    sd1, sd2 = model.state_dict(), m2.state_dict()
    assert sd1.keys() == sd2.keys()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k


def test_backbone_load_param_prefix_rules(tmp_path):
    """ResNet.load_param / ResNet_IBN.load_param (resnet.py:135-154, resnet_ibn_a.py:143-162): 'backbone.base.' and
    'base.' prefixes are stripped, fc / classifier / bottleneck entries skipped, a wrapping {'state_dict': ...} opened."""
    import torch
    from centroids_reid_amd import backbone as bb
    src = bb.ResNet(last_stride=1)
    with torch.no_grad():
        for p in src.parameters():
            p.normal_()
    sd = src.state_dict()
    for prefix, wrap in (("backbone.base.", True), ("base.", False), ("", False)):
        d = {prefix + k: v.clone() for k, v in sd.items()}
        d[prefix + "fc.weight"] = torch.zeros(3)                 # must be ignored
        d["classifier.weight"] = torch.zeros(2)
        path = tmp_path / f"w_{len(prefix)}.pth"
        torch.save({"state_dict": d} if wrap else d, path)
        dst = bb.ResNet(last_stride=1)
        dst.load_param(str(path))
        for k, v in dst.state_dict().items():
            assert torch.equal(v, sd[k]), (prefix, k)
    isrc = bb.resnet50_ibn_a(1)
    isd = {("base." + k): v.clone() for k, v in isrc.state_dict().items()}
    isd["classifier.weight"] = torch.zeros(2)
    ipath = tmp_path / "ibn.pth"
    torch.save(isd, ipath)
    idst = bb.resnet50_ibn_a(1)
    with torch.no_grad():
        for p in idst.parameters():
            p.add_(1.0)
    idst.load_param(str(ipath))
    for k, v in idst.state_dict().items():
        if not k.startswith("fc."):
            assert torch.equal(v, isrc.state_dict()[k]), k


def test_config_tree_has_exactly_the_reference_keys_and_defaults():
    """north_star: "config/defaults.py keys ... stay intact".  tests/golden/config_keys.json is the reference's own tree
    (config/defaults.py:13-181, recorded by tools/gen_golden.py config): same dotted key set, same defaults."""
    import json
    import os
    from centroids_reid_amd.config import get_cfg_defaults
    here = os.path.dirname(os.path.abspath(__file__))
    ref = json.load(open(os.path.join(here, "golden", "config_keys.json")))["keys"]

    def flat(node, prefix, out):
        for k, v in node.items():
            if isinstance(v, dict):
                flat(v, prefix + k + ".", out)
            else:
                out[prefix + k] = list(v) if isinstance(v, tuple) else v
        return out
    own = flat(get_cfg_defaults(), "", {})
    assert sorted(own) == sorted(ref), (sorted(set(ref) - set(own)), sorted(set(own) - set(ref)))
    diff = {k: (own[k], ref[k]) for k in ref if own[k] != ref[k]}
    assert not diff, diff


def test_learning_rate_schedule_matches_the_reference_recording(golden):
    """solver/build.py:50-63 + the warm-up of train_ctl_model.py:41-49: the learning rate Adam steps with, epoch by epoch, for both
    scheduler names with and without warm-up, against `tools/gen_golden.py autocast` -> lr_schedule.npz (the reference's optimiser,
    scheduler and warm-up lines run for 120 epochs).  Here: this package's FusedAdam / CenterSGD param_groups under the same torch
    scheduler, the warm-up as CTLModel.forward_backward applies it."""
    import warnings
    import torch
    from centroids_reid_amd.config import get_cfg_defaults
    from centroids_reid_amd.solver import build_optimizer, build_scheduler
    g = golden("lr_schedule")
    for sched in ("multistep_lr", "cosine_annealing"):
        for warm in (True, False):
            cfg = get_cfg_defaults()
            cfg.SOLVER.LR_SCHEDULER_NAME = sched
            cfg.SOLVER.USE_WARMUP_LR = warm
            if sched == "cosine_annealing":
                cfg.SOLVER.MIN_LR = 1e-7
            w = torch.nn.Parameter(torch.zeros(4)); c = torch.nn.Parameter(torch.zeros(4))
            opts = build_optimizer([("w", w), ("center_loss.centers", c)], cfg)
            sch = build_scheduler(opts[0], cfg)
            used, center = [], []
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                 # "scheduler.step() before optimizer.step()": no device here to step on
                for epoch in range(int(cfg.SOLVER.MAX_EPOCHS)):
                    if cfg.SOLVER.USE_WARMUP_LR and epoch < cfg.SOLVER.WARMUP_EPOCHS:
                        lr_scale = min(1.0, float(epoch + 1) / float(cfg.SOLVER.WARMUP_EPOCHS))
                        for pg in opts[0].param_groups:
                            pg["lr"] = lr_scale * cfg.SOLVER.BASE_LR
                    used.append(opts[0].param_groups[0]["lr"]); center.append(opts[1].param_groups[0]["lr"])
                    sch.step()
            key = f"{sched}_{'warm' if warm else 'nowarm'}"
            np.testing.assert_allclose(np.array(used), g[key], rtol=1e-12, atol=0, err_msg=key)
            np.testing.assert_allclose(np.array(center), g[key + "_center"], rtol=0, atol=0, err_msg=key)
