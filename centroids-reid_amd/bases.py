"""modelling/bases.py:52-393 ModelBase -- the LightningModule surface of the reference, kept as the
drop-in boundary (same constructor, attribute names, hook names and return values) with every
arithmetic step routed to the HIP kernels.  pytorch-lightning is optional: when it is importable the
class derives from pl.LightningModule, otherwise from nn.Module with a minimal trainer stand-in
(`self.trainer.current_epoch`, `optimizers()`, `manual_backward()`), which is all the hot path uses.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .baseline import Baseline
from .config import CfgNode
from .losses import BatchNorm1d, CenterLoss, CrossEntropyLabelSmooth, Linear, TripletLoss
from .reid_metric import R1_mAP
from .solver import build_optimizer, build_scheduler

try:  # optional
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


class AttributeDict(CfgNode):
    pass


def _to_attr(d):
    if isinstance(d, dict):
        out = AttributeDict()
        for k, v in d.items():
            out[k] = _to_attr(v)
        return out
    return d


def _plain_dict(d):
    """dict subclasses (AttributeDict / CfgNode stand-ins) -> plain nested dicts."""
    if isinstance(d, dict):
        return {k: _plain_dict(v) for k, v in d.items()}
    return d


def load_checkpoint_file(path, map_location="cpu"):
    """torch.load for the reference's pytorch-lightning 1.1.4 checkpoints (utils/misc.py:80-93): the pickled
    `hyper_parameters` reference `pytorch_lightning.utilities.parsing.AttributeDict` and `yacs.config.CfgNode`
    (plain dict subclasses).  When those packages are absent, stand-in dict subclasses are registered under the
    pickled module paths for the duration of the load, so the file opens without either dependency."""
    import importlib
    import sys
    import types
    wanted = {"pytorch_lightning.utilities.parsing": ("AttributeDict",), "pytorch_lightning.utilities": ("AttributeDict",),
              "yacs.config": ("CfgNode",)}
    added = []
    for mod_name, classes in wanted.items():
        try:
            importlib.import_module(mod_name)
            continue
        except Exception:  # noqa: BLE001
            pass
        parts = mod_name.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                added.append(name)
        for c in classes:
            if not hasattr(sys.modules[mod_name], c):
                setattr(sys.modules[mod_name], c, type(c, (AttributeDict,), {"__module__": mod_name}))
    try:
        return torch.load(path, map_location=map_location, weights_only=False)
    finally:
        for name in added:
            sys.modules.pop(name, None)


class ModelBase(_Base):
    def __init__(self, cfg=None, test_dataloader=None, compute_dtype=None, **kwargs):
        super().__init__()
        if cfg is None:
            hparams = {**kwargs}
        elif isinstance(cfg, dict):
            hparams = {**cfg, **kwargs}
            if cfg["TEST"]["ONLY_TEST"]:
                hparams = {**kwargs, **cfg}            # modelling/bases.py:59-62
        else:
            raise TypeError("cfg must be a dict-like config tree")
        hp = _to_attr(hparams)
        if pl is not None:
            self.save_hyperparameters(hp)
        else:
            self.hparams = hp
        if test_dataloader is not None:
            self.test_dataloader = test_dataloader

        self.backbone = Baseline(self.hparams, compute_dtype=compute_dtype)
        self.backbone.return_base_out = False            # this module only reads global_feat (modelling/bases.py:171)
        self.contrastive_loss = TripletLoss(self.hparams.SOLVER.MARGIN, self.hparams.SOLVER.DISTANCE_FUNC)
        d_model = self.hparams.MODEL.BACKBONE_EMB_SIZE
        self.xent = CrossEntropyLabelSmooth(num_classes=self.hparams.num_classes)
        self.center_loss = CenterLoss(num_classes=self.hparams.num_classes, feat_dim=d_model)
        self.center_loss_weight = self.hparams.SOLVER.CENTER_LOSS_WEIGHT
        self.bn = BatchNorm1d(d_model)
        self.bn.bias.requires_grad_(False)               # modelling/bases.py:84
        self.fc_query = Linear(d_model, self.hparams.num_classes, bias=False)
        self.losses_names = ["query_xent", "query_triplet", "query_center"]
        self.losses_dict = {n: [] for n in self.losses_names}
        if pl is None:
            self.trainer = SimpleNamespace(current_epoch=0, global_rank=0, local_rank=0, logger=None,
                                           train_dataloader=None)
            self._optimizers = None

    # ------------------------------------------------------------------ optimiser plumbing
    @staticmethod
    def _calculate_centroids(vecs, dim=1):
        length = vecs.shape[dim]
        return torch.sum(vecs, dim) / length

    def configure_optimizers(self):
        optimizers_list = build_optimizer(self.named_parameters(), self.hparams)
        self.lr_scheduler = build_scheduler(optimizers_list[0], self.hparams)
        if getattr(self.backbone, "compute_dtype", None) == torch.float16 and optimizers_list[0].flat.is_cuda:
            # the reference's precision=16 (utils/misc.py:111): f16 backbone + dynamic loss scale, state on the device
            from .solver import LossScaler
            self.loss_scaler = LossScaler(optimizers_list[0].flat.device)
            optimizers_list[0].attach_scaler(self.loss_scaler)
            optimizers_list[1].scaler = self.loss_scaler
            self.backbone.loss_scaler = self.loss_scaler
        if pl is None:
            self._optimizers = optimizers_list
        return optimizers_list, self.lr_scheduler

    if pl is None:
        def optimizers(self, use_pl_optimizer=True):
            if self._optimizers is None:
                self.configure_optimizers()
            return self._optimizers

        def manual_backward(self, loss, optimizer=None):
            loss.backward()

        @property
        def current_epoch(self):
            return self.trainer.current_epoch

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, optimizer_idx=None, optimizer_closure=None,
                       on_tpu=False, using_native_amp=False, using_lbfgs=False, **kwargs):
        """modelling/bases.py:102-133: linear learning-rate warm-up over SOLVER.WARMUP_EPOCHS applied to the
        optimiser being stepped, then the trainer's default step (closure first, as PL-1.1.4 does)."""
        hp = self.hparams
        if hp.SOLVER.USE_WARMUP_LR and epoch is not None and epoch < hp.SOLVER.WARMUP_EPOCHS:
            lr_scale = min(1.0, float(epoch + 1) / float(hp.SOLVER.WARMUP_EPOCHS))
            for pg in optimizer.param_groups:
                pg["lr"] = lr_scale * hp.SOLVER.BASE_LR
        if pl is not None:
            return super().optimizer_step(epoch=epoch, batch_idx=batch_idx, optimizer=optimizer,
                                          optimizer_idx=optimizer_idx, optimizer_closure=optimizer_closure, on_tpu=on_tpu,
                                          using_native_amp=using_native_amp, using_lbfgs=using_lbfgs, **kwargs)
        if optimizer_closure is not None:
            optimizer_closure()
        optimizer.step()

    # ------------------------------------------------------------------ checkpoint IO
    def checkpoint_dict(self, epoch=0, global_step=0):
        """The pytorch-lightning 1.1.4 checkpoint layout the reference's callbacks write
        (utils/misc.py:80-93, callbacks/chechpointer_callback.py:56-74): state_dict (325 keys for R50-CTL),
        hyper_parameters, optimizer_states, lr_schedulers, epoch, global_step, callbacks."""
        def plain(d):
            return {k: plain(v) if isinstance(v, dict) else v for k, v in d.items()}
        opts = self._optimizers if pl is None else None
        extra = {}
        if getattr(self, "loss_scaler", None) is not None:
            # precision=16 runs: pytorch-lightning 1.1.4 stores GradScaler.state_dict() under this key (same fields)
            extra["native_amp_scaling_state"] = self.loss_scaler.state_dict()
        return {
            **extra,
            "epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "1.1.4",
            "state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
            "hparams_name": "kwargs", "hyper_parameters": plain(dict(self.hparams)),
            "optimizer_states": [o.state_dict() for o in (opts or [])],
            "lr_schedulers": [self.lr_scheduler.state_dict()] if hasattr(self, "lr_scheduler") else [],
            "callbacks": {},
        }

    def save_checkpoint(self, path, epoch=0, global_step=0):
        torch.save(self.checkpoint_dict(epoch, global_step), path)

    def load_training_state(self, ckpt):
        """Resume: what pytorch-lightning 1.1.4 restores next to the weights (checkpoint_connector.restore_training_state) --
        `optimizer_states` (Adam moments + step count into the flat buffers, FusedAdam.load_state_dict), `lr_schedulers`, and for
        precision=16 runs `native_amp_scaling_state` (GradScaler scale + growth tracker -> the device-resident LossScaler).
        Call after configure_optimizers(); `ckpt` is the dict load_checkpoint_file() returns (or a path)."""
        if not isinstance(ckpt, dict):
            ckpt = load_checkpoint_file(ckpt, "cpu")
        opts = getattr(self, "_optimizers", None)
        if opts is None:
            raise RuntimeError("load_training_state() needs configure_optimizers() first")
        for o, st in zip(opts, ckpt.get("optimizer_states", [])):
            o.load_state_dict(st)
        for sched, st in zip([self.lr_scheduler] if hasattr(self, "lr_scheduler") else [], ckpt.get("lr_schedulers", [])):
            sched.load_state_dict(st)
        amp = ckpt.get("native_amp_scaling_state")
        if amp is not None and getattr(self, "loss_scaler", None) is not None:
            self.loss_scaler.load_state_dict(amp)
        return ckpt.get("epoch", 0), ckpt.get("global_step", 0)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", strict=True, **overrides):
        """pl.LightningModule.load_from_checkpoint as the reference uses it (utils/misc.py:128-147,
        inference/get_similar.py:84): rebuild the module from `hyper_parameters`, then load `state_dict`.
        Checkpoints written by the reference pickle `hyper_parameters` as pytorch-lightning `AttributeDict` /
        yacs `CfgNode` objects; neither package needs to be installed to read them (see load_checkpoint_file)."""
        ckpt = load_checkpoint_file(checkpoint_path, map_location)
        hp = _to_attr({**_plain_dict(ckpt["hyper_parameters"]), **overrides})
        hp["MODEL"]["PRETRAINED"] = False            # weights come from the checkpoint, not from ImageNet
        model = cls(cfg=None, **hp)
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        return model

    def training_step(self, batch, batch_idx, opt_idx=None):
        raise NotImplementedError("A used model should have its own training_step method implemented")

    def training_epoch_end(self, outputs):
        """modelling/bases.py:140-167 (logging; the barrier is the trainer's business)."""
        sampler = getattr(getattr(self.trainer, "train_dataloader", None), "sampler", None)
        if sampler is not None and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(self.current_epoch + 1)
        lr = self.lr_scheduler.get_last_lr()[0]
        loss = torch.stack([x.pop("loss").detach().float() for x in outputs]).mean().cpu()
        log_data = {"epoch_train_loss": float(loss), "lr": lr}
        for k_out, k_in in (("epoch_dist_ap", "step_dist_ap"), ("epoch_dist_an", "step_dist_an"),
                            ("l2_mean_centroid", "l2_mean_centroid")):
            log_data[k_out] = float(np.mean([float(x["other"].pop(k_in)) for x in outputs]))
        if hasattr(self, "losses_dict"):
            for name, vals in self.losses_dict.items():
                log_data[name] = float(np.mean([float(v) for v in vals])) if vals else float("nan")
                self.losses_dict[name] = []
        logger = getattr(self.trainer, "logger", None)
        if logger is not None:
            logger.log_metrics(log_data, step=self.trainer.current_epoch)
        return log_data

    # ------------------------------------------------------------------ validation
    def validation_step(self, batch, batch_idx):
        """modelling/bases.py:169-177: eval-mode backbone + BNNeck embedding."""
        self.backbone.eval()
        self.bn.eval()
        x, class_labels, camid, idx = batch
        with torch.no_grad():
            _, emb = self.backbone(x)
            emb = self.bn(emb)
        return {"emb": emb, "labels": class_labels, "camid": camid, "idx": idx}

    def validation_create_centroids(self, embeddings, labels, camids, respect_camids=False):
        """modelling/bases.py:179-262 (respect_camids=False): gallery -> per-PID mean (device kernel);
        returns (embeddings [nq + n_centroids, D], labels, camids) with the reference's dummy camids
        (including its nq-longer-than-needed camid vector, bases.py:255-260)."""
        num_query = self.hparams.num_query
        labels = np.asarray(labels)
        emb = embeddings.float().contiguous()
        L.require_gpu(emb)
        lq, lg = labels[:num_query], labels[num_query:]
        groups, cent_labels, cent_cams = [], [], []
        if respect_camids:
            # modelling/bases.py:205-236.  For each gallery PID and each distinct camera of its QUERIES: one
            # centroid of the gallery rows from other cameras, de-duplicated by camera set.  The reference looks
            # the gallery cameras up as camids[inds] with gallery-relative indices on the full vector (:214);
            # that quirk is kept so results match.
            camids = np.asarray(camids)
            for u in np.unique(lg):
                inds = np.nonzero(lg == u)[0]
                cams_g = camids[inds]
                seen = set()
                for cur in np.unique(camids[np.nonzero(lq == u)[0]]):
                    sel = np.nonzero(cams_g != cur)[0]
                    if len(sel) == 0:
                        continue
                    used = tuple(sorted(np.unique(cams_g[cams_g != cur]).tolist()))
                    if used in seen:
                        continue
                    seen.add(used)
                    groups.append(inds[sel]); cent_labels.append(u); cent_cams.append(list(used))
        else:
            uniq, inverse = np.unique(lg, return_inverse=True)
            order_all = np.argsort(inverse, kind="stable")        # gallery rows grouped by PID, original order kept
            counts = np.bincount(inverse, minlength=len(uniq))
            bounds = np.concatenate([[0], np.cumsum(counts)])
            groups = [order_all[bounds[i]:bounds[i + 1]] for i in range(len(uniq))]
            cent_labels = list(uniq)
        order = np.concatenate(groups).astype(np.int64) + num_query
        offsets = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int64)
        dev = emb.device
        order_t = torch.as_tensor(order, device=dev)
        off_t = torch.as_tensor(offsets, device=dev)
        cents = torch.empty((len(groups), emb.shape[1]), dtype=torch.float32, device=dev)
        L.check(L.lib().creid_gather_mean_rows(L.ptr(emb), L.ptr(order_t), L.ptr(off_t), len(groups), emb.shape[1],
                                               L.ptr(cents), L.stream()), "creid_gather_mean_rows")
        out = torch.cat((emb[:num_query], cents), dim=0)
        out_labels = np.hstack((lq, np.asarray(cent_labels)))
        if respect_camids:
            out_camids = [[c] for c in camids[:num_query]] + cent_cams             # :250-253
        else:
            out_camids = np.hstack((np.zeros_like(lq), np.ones_like(out_labels)))   # :255-260 (quirk kept)
        return out, out_labels, out_camids

    def get_val_metrics(self, embeddings, labels, camids):
        """modelling/bases.py:264-297."""
        self.r1_map_func = R1_mAP(pl_module=self, num_query=self.hparams.num_query,
                                  feat_norm=self.hparams.TEST.FEAT_NORM, streamed=True)   # only the metrics are used
        respect_camids = bool(self.hparams.MODEL.KEEP_CAMID_CENTROIDS and self.hparams.MODEL.USE_CENTROIDS)
        cmc, mAP, all_topk = self.r1_map_func.compute(feats=embeddings.float(), pids=labels, camids=camids,
                                                      respect_camids=respect_camids)
        topks = {}
        for top_k, kk in zip(all_topk, [1, 5, 10, 20, 50]):
            print("top-k, Rank-{:<3}:{:.1%}".format(kk, top_k))
            topks[f"Top-{kk}"] = top_k
        print(f"mAP: {mAP}")
        log_data = {"mAP": mAP, **topks}
        logger = getattr(self.trainer, "logger", None)
        if logger is not None:
            logger.log_metrics(log_data, step=self.trainer.current_epoch)
        self.last_val_metrics = log_data
        return log_data

    def validation_epoch_end(self, outputs):
        """modelling/bases.py:299-318 -- embeddings stay on the device (the reference moves them to CPU)."""
        embeddings = torch.cat([x.pop("emb") for x in outputs]).detach()
        labels = torch.cat([x.pop("labels") for x in outputs]).detach().cpu().numpy()
        camids = torch.cat([x.pop("camid") for x in outputs]).cpu().detach().numpy()
        del outputs
        if self.hparams.MODEL.USE_CENTROIDS:
            print("Evaluation is done using centroids")
            embeddings, labels, camids = self.validation_create_centroids(
                embeddings, labels, camids, respect_camids=self.hparams.MODEL.KEEP_CAMID_CENTROIDS)
        metrics = self.get_val_metrics(embeddings, labels, camids)
        if pl is None and self.training:
            # validation_step put backbone / BNNeck in eval mode (modelling/bases.py:170-171); the reference relies on
            # the PL trainer to switch the module back for the next training epoch -- without a trainer do it here
            self.backbone.train()
            self.bn.train()
        return metrics

    @staticmethod
    def create_masks_train(class_labels):
        """modelling/bases.py:359-384 -> (masks bool [K_max, B], per-PID index lists in first-appearance order).
        masks[r, j] is False iff sample j is the r-th occurrence of its PID (the query of round r).  A PID with
        c < K_max occurrences additionally gets, in rounds r >= c, the index range [cs, cs + c) cleared, where
        cs is the CUMULATIVE COUNT of the PIDs before it (for the first PID the reference reads cumsum[-1], i.e.
        an empty range) -- that range is the PID's own block only when the batch is PID-contiguous."""
        labels = np.asarray(class_labels.detach().cpu().numpy())
        B = labels.shape[0]
        _, first, inverse, counts = np.unique(labels, return_index=True, return_inverse=True, return_counts=True)
        by_first = np.argsort(first, kind="stable")                  # PID groups in order of first appearance
        group_of_unique = np.empty_like(by_first)
        group_of_unique[by_first] = np.arange(len(by_first))
        group = group_of_unique[inverse.reshape(-1)]                 # group id of every sample
        cnt = counts[by_first]
        perm = np.argsort(group, kind="stable")                      # samples grouped, batch order kept inside
        ends = np.cumsum(cnt)
        starts = ends - cnt
        occurrence = np.empty(B, np.int64)
        occurrence[perm] = np.arange(B) - np.repeat(starts, cnt)
        k_max = int(cnt.max())
        masks = np.ones((k_max, B), dtype=bool)
        masks[occurrence, np.arange(B)] = False
        for g in np.nonzero(cnt < k_max)[0]:
            lo = int(ends[g - 1])                                    # g == 0 wraps to the total (reference quirk)
            masks[cnt[g]:, lo:lo + int(cnt[g])] = False
        index_lists = [perm[starts[g]:ends[g]].tolist() for g in range(len(cnt))]
        return torch.from_numpy(masks), index_lists

    def test_step(self, batch, batch_idx):
        return self.validation_step(batch, batch_idx)

    def test_epoch_end(self, outputs):
        return self.validation_epoch_end(outputs)
