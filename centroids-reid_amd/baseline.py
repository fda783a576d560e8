"""modelling/baseline.py:44-107 Baseline on the HIP backbone engine.

`Baseline(cfg).forward(x) -> (base_out, global_feat)`; `self.base` holds the backbone parameters
under the reference's names (state_dict keys `base.conv1.weight`, ...).  `compute_dtype` selects the
activation / MFMA input type: torch.bfloat16 (throughput mode; the reference's AMP analogue when
cfg.USE_MIXED_PRECISION) or torch.float32 (parity mode, exact-f32 MFMA)."""
from __future__ import annotations

import torch
from torch import nn

from . import backbone as bb


class Baseline(nn.Module):
    in_planes = 2048

    def __init__(self, cfg, compute_dtype=None):
        super().__init__()
        last_stride = cfg.MODEL.LAST_STRIDE
        model_name = cfg.MODEL.NAME
        self.use_mixed_precision = cfg.USE_MIXED_PRECISION
        # modelling/baseline.py:56-81: resnet50 / 101 / 152 and resnet50_ibn_a / resnet101_ibn_a (Bottleneck networks, in_planes
        # 2048); resnet18 / resnet34 (BasicBlock, in_planes 512; round 6)
        self.base = bb.build_backbone(model_name, last_stride)
        self.in_planes = self.base.out_channels
        self.model_name = model_name
        if cfg.MODEL.PRETRAINED and not cfg.MODEL.RESUME_TRAINING and not cfg.TEST.ONLY_TEST:
            self.base.load_param(cfg.MODEL.PRETRAIN_PATH)      # modelling/baseline.py:84-87
            print("Loading pretrained ImageNet model......")
        self.compute_dtype = compute_dtype or (torch.bfloat16 if cfg.USE_MIXED_PRECISION else torch.float32)
        # base_out = the reference's NCHW fp32 feature map (modelling/baseline.py:91-96).  A stand-alone Baseline returns it like the
        # reference does; ModelBase / CTLModel, which never consume it (`_, features = self.backbone(x)`, modelling/bases.py:171,
        # train_ctl_model.py:44), switch it off and save the 67 MB layout pass per batch.  It is a detached copy: gradients flow
        # through global_feat only.
        self.return_base_out = True
        self._engine = None
        self.loss_scaler = None           # f16 training: solver.LossScaler (ModelBase.configure_optimizers attaches it)

    @property
    def engine(self):
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            self._engine = bb.BackboneEngine(self.base, self.compute_dtype)
        self._engine.loss_scaler = self.loss_scaler
        return self._engine

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.fold_counters()
        return super().state_dict(*args, **kwargs)

    def forward(self, x):
        eng = self.engine
        if isinstance(x, torch.Tensor):                 # (a transforms.StemOperand passes through: already the stem's layout)
            x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            base_out, feat = bb._BackboneFn.apply(x, self.base.conv1.weight, eng, self.return_base_out)
            if base_out.numel() == 0:
                base_out = None
        else:
            base_out, feat = eng.forward(x, self.training, self.return_base_out)
        return base_out, feat

    def load_param(self, trained_path, load_specific=None):
        param_dict = torch.load(trained_path, map_location="cpu", weights_only=False)
        for i in param_dict:
            if load_specific is not None:
                if load_specific in i:
                    self.state_dict()[i].copy_(param_dict[i])
            else:
                if "classifier" in i:
                    continue
                self.state_dict()[i].copy_(param_dict[i])
        self.engine.weights_dirty = True
