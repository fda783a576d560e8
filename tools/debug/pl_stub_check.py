"""Executes the `pytorch_lightning` branch of centroids-reid_amd (bases.py derives from pl.LightningModule when the package
imports; train_ctl_model._raw_optimizers unwraps LightningOptimizer) against the reference's own training_step recordings,
with the stand-in package of tools/debug/pl_stub on the path.  Prints PL_BRANCH_OK.
    python tools/debug/pl_stub_check.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "debug", "pl_stub"))
sys.path.insert(0, ROOT)
import pytorch_lightning as pl  # noqa: E402
from centroids_reid_amd import bases  # noqa: E402
from centroids_reid_amd.train_ctl_model import CTLModel  # noqa: E402
from centroids_reid_amd.config import get_cfg_defaults  # noqa: E402

assert bases.pl is pl and issubclass(CTLModel, pl.LightningModule), "the PL branch was not taken"


class EngineStub:
    def __init__(self, owner):
        self.owner = owner
        self.weights_dirty = False

    def forward(self, x, training, want_base_out=False):
        return None, self.owner.feats.detach() * 1.0

    def backward(self, dfeat):
        self.owner.feats.grad.add_(dfeat)


class FeatStub(torch.nn.Module):
    def __init__(self, feats, with_engine):
        super().__init__()
        self.feats = torch.nn.Parameter(feats.clone())
        if with_engine:
            self.engine = EngineStub(self)

    def forward(self, x):
        return None, self.feats * 1.0


ok = True
for name in ("heads_p16k4_d128", "heads_p16k4_d128_fake1"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for fused in (False, True):
        P, K, C = int(g["P"]), int(g["K"]), int(g["C"])
        D = g["feats"].shape[1]
        cfg = get_cfg_defaults()
        cfg.MODEL.PRETRAINED = False; cfg.MODEL.BACKBONE_EMB_SIZE = D; cfg.DATALOADER.NUM_INSTANCE = K
        cfg.SOLVER.MARGIN = float(g["margin"]); cfg.USE_MIXED_PRECISION = False
        model = CTLModel(cfg, num_classes=C, num_query=0)
        model.backbone = FeatStub(torch.from_numpy(g["feats"]), fused)
        with torch.no_grad():
            model.center_loss.centers.copy_(torch.from_numpy(g["centers0"]))
            model.fc_query.weight.copy_(torch.from_numpy(g["fc0"]))
            model.bn.weight.copy_(torch.from_numpy(g["bn_w0"]))
        model = model.cuda().train()
        opts = model.optimizers()                                    # LightningOptimizer wrappers
        assert all(type(o).__name__ == "LightningOptimizer" for o in opts)
        batch = (torch.zeros(P * K, 3, 8, 4, device="cuda"), torch.from_numpy(g["labels"]).cuda(),
                 torch.zeros(P * K, dtype=torch.int64), torch.from_numpy(g["is_real"]))
        nsteps = 2 if "s1_loss_total" in g else 1
        for s in range(nsteps):
            out = model.training_step(batch, s)
            e = abs(float(out["loss"]) - float(g[f"s{s}_loss_total"]))
            # the center update only matches if grad_mul reached the WRAPPED optimizer (train_ctl_model.py:157-159)
            ec = float(np.abs(model.center_loss.centers.detach().cpu().numpy() - g[f"s{s}_centers_after"]).max())
            ef = float(np.abs(model.fc_query.weight.detach().cpu().numpy() - g[f"s{s}_fc_after"]).max())
            good = e < 3e-5 and ec < 1e-4 and ef < 1e-5
            print(name, "fused" if fused else "autograd", "step", s, "loss err", e, "centers err", ec, "fc err", ef, "OK" if good else "MISMATCH")
            ok &= good
# optimizer_step goes through the LightningModule's own implementation on this branch (modelling/bases.py:102-133)
cfg = get_cfg_defaults(); cfg.MODEL.PRETRAINED = False
m = CTLModel(cfg, num_classes=10, num_query=0).cuda()
raw = m.optimizers(use_pl_optimizer=False)[0]
m.optimizer_step(epoch=0, batch_idx=0, optimizer=raw, optimizer_idx=0)
lr = raw.param_groups[0]["lr"]
ok &= abs(lr - cfg.SOLVER.BASE_LR * (1.0 / cfg.SOLVER.WARMUP_EPOCHS)) < 1e-12 if cfg.SOLVER.USE_WARMUP_LR else True
print("warm-up lr after optimizer_step(epoch=0):", lr)
print("PL_BRANCH_OK" if ok else "PL_BRANCH_MISMATCH")
