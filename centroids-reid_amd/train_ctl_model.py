"""train_ctl_model.py:27-179 CTLModel -- the centroid-triplet training step on the HIP kernels.

Same class name, constructor, `losses_names`, `training_step(batch, batch_idx, optimizer_idx=None)`
signature and return value `{"loss": ..., "other": {"step_dist_ap", "step_dist_an",
"l2_mean_centroid"}}` as the reference.  Differences that are deliberate and documented in DESIGN.md:
  * the batch is PID-contiguous [P, K] (the reference assumes the same at :80), so P = B / K is known
    on the host and the round masks come from `isReal` alone -- no D2H sync for np.unique / masks;
  * a centroid is dropped from a round iff it has no real member (the reference tests
    |centroid|_1 > 1e-7, which differs only for an exactly-zero mean);
  * the seven per-step `float()` host syncs are deferred: the logged values are 0-dim device tensors
    (they still convert with float()).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .bases import ModelBase


class CTLModel(ModelBase):
    def __init__(self, cfg=None, **kwargs):
        super().__init__(cfg, **kwargs)
        self.losses_names = ["query_xent", "query_triplet", "query_center", "centroid_triplet"]
        self.losses_dict = {n: [] for n in self.losses_names}
        self.grad_sync = None          # optional callable(model) run between backward and the optimiser steps
        # all-real batches on the HIP backbone take a hand-scheduled head pass (same kernels, no autograd tape:
        # ~45 launches instead of ~130); CREID_FUSED_HEADS=0 keeps everything on the autograd path
        self.fused_heads = os.environ.get("CREID_FUSED_HEADS", "1") == "1"
        # ... and that head pass as SIX multi-role launches (creid_ctl_heads_fused: the four independent chains of the heads side
        # by side, the last launch also doing the global-average-pool backward); CREID_HEADS_ONE_CALL=0: the ~23 separate launches
        self.heads_one_call = os.environ.get("CREID_HEADS_ONE_CALL", "1") == "1"

    def training_step(self, batch, batch_idx, optimizer_idx=None):
        """train_ctl_model.py:38-179 = forward_backward (everything up to manual_backward) -> optional
        data-parallel gradient sync -> apply_optimizers.  The two halves are separately callable so that a
        launcher can capture each into a hipGraph and run the RCCL all-reduce between them."""
        out = self.forward_backward(batch, batch_idx)
        if self.grad_sync is not None:
            self.grad_sync(self)                                               # data-parallel all-reduce (RCCL)
        self.apply_optimizers()
        return out

    def check_lonely_identities(self):
        """Steps driven by a DEVICE isReal mask cannot raise inside the step (no host sync); the kernel counts every real
        instance that had no real partner, and this raises the reference's error for them -- once per epoch from
        training_epoch_end, or whenever the caller wants to pay the 4-byte read-back."""
        lonely = getattr(self, "_lonely_dev", None)
        if lonely is None:
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # every rank must take the same decision: a rank that raised alone would leave the others hanging in their next
            # collective (each rank sees only its own batches)
            tot = lonely.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            n = int(tot.item())
        else:
            n = int(lonely.item())
        if n:
            lonely.zero_()
            raise RuntimeError(f"query/centroid count mismatch in a centroid round: {n} real instance(s) without a real partner "
                               "in the steps since the last check (the reference fails in labels.expand at "
                               "losses/triplet_loss.py:88)")

    def training_epoch_end(self, outputs):
        self.check_lonely_identities()
        return super().training_epoch_end(outputs)

    def _raw_optimizers(self):
        """(Adam, center SGD) as the objects whose step() actually runs: under pytorch-lightning
        `self.optimizers()` returns LightningOptimizer wrappers, and attributes set on a wrapper (grad_mul,
        grad_scale) would never reach the wrapped optimizer."""
        opt, opt_center = self.optimizers(use_pl_optimizer=True)
        return getattr(opt, "_optimizer", opt), getattr(opt_center, "_optimizer", opt_center)

    def apply_optimizers(self):
        hp = self.hparams
        opt, opt_center = self._raw_optimizers()
        opt.step()                                                             # :155
        eng = getattr(self.backbone, "_engine", None)
        if eng is not None:
            eng.weights_dirty = True
        opt_center.grad_mul = 1.0 / hp.SOLVER.CENTER_LOSS_WEIGHT               # :157-158 (fused into the step)
        opt_center.step()                                                      # :159
        scaler = getattr(self, "loss_scaler", None)
        if scaler is not None:
            scaler.update()                                                    # f16: GradScaler.update() after both steps

    def forward_backward(self, batch, batch_idx=0):
        hp = self.hparams
        opt, opt_center = self._raw_optimizers()
        if hp.SOLVER.USE_WARMUP_LR:                                           # :41-49
            if self.trainer.current_epoch < hp.SOLVER.WARMUP_EPOCHS:
                lr_scale = min(1.0, float(self.trainer.current_epoch + 1) / float(hp.SOLVER.WARMUP_EPOCHS))
                for pg in opt.param_groups:
                    pg["lr"] = lr_scale * hp.SOLVER.BASE_LR
        if not (self.backbone.training and self.bn.training):
            raise RuntimeError("training_step with backbone / BNNeck in eval mode (validation_step leaves them there, "
                               "modelling/bases.py:170-171): call model.train() first, as the PL trainer does")
        if not getattr(opt_center, "grad_in_adam_tail", False):
            opt_center.zero_grad()        # (in the tail of Adam's flat gradient buffer: the one fill below covers it)
        opt.zero_grad()

        x, class_labels, camid, isReal = batch
        K = hp.DATALOADER.NUM_INSTANCE
        B = x.shape[0]
        assert B % K == 0, "batch must be PID-contiguous [P, K] (datasets/bases.py:447-455 collate)"
        P = B // K
        dev = x.device
        # hand-scheduled heads (same kernels, no autograd tape).  K <= 16: creid_loo_emb_bwd keeps one register slot per
        # instance of a pid.  isReal on the HOST: all-real batches take the unmasked schedule, batches with padded samples
        # (isReal = False, datasets/bases.py:346-406) the masked one.  isReal on the DEVICE (what Lightning hands over once it has
        # moved the batch): ALWAYS the masked schedule, driven by the device mask -- no host synchronisation for any pattern of
        # fakes (hipGraph-capturable, bench.py's "fake_mix" line).  One divergence from the reference on that path, accepted for
        # the missing sync: an identity with exactly ONE real instance makes the reference fail in labels.expand
        # (losses/triplet_loss.py:88; the host path below raises the same way); the device path cannot look at the mask, so the
        # lonely row drops out of its round (`exists = qreal && cnt > 0` in creid_loo_emb_fwd_rows), the kernel COUNTS it on the
        # device, and the same RuntimeError is raised late: at training_epoch_end (or by check_lonely_identities()).
        # The reference's own sampler never produces such a batch (PK batches pad whole instances of an identity that has
        # at least two real ones, datasets/bases.py:374-395).
        fused_ok = (self.fused_heads and P >= 2 and 2 <= K <= 16 and x.is_cuda and hasattr(self.backbone, "engine")
                    and self.backbone.training and self.contrastive_loss.margin is not None
                    and self.contrastive_loss.dist_name == "euclidean")
        if fused_ok and isinstance(isReal, torch.Tensor) and isReal.is_cuda:
            real = isReal if isReal.dtype == torch.uint8 else isReal.to(torch.uint8)
            return self._forward_backward_fused(x, class_labels.to(dev, non_blocking=True), P, K, real=real.contiguous())
        ir_host = np.asarray(isReal.cpu() if isinstance(isReal, torch.Tensor) else isReal, dtype=bool)
        all_real = bool(ir_host.all())
        if fused_ok and not all_real:
            ir2 = ir_host.reshape(P, K)
            lonely = ir2 & (ir2.sum(1, keepdims=True) == 1)                  # a real instance without a real partner
            if lonely.any():
                raise RuntimeError("query/centroid count mismatch in a centroid round (the reference fails in "
                                   "labels.expand at losses/triplet_loss.py:88)")
            real = torch.as_tensor(ir_host.astype(np.uint8)).to(dev)
            return self._forward_backward_fused(x, class_labels.to(dev, non_blocking=True), P, K, real=real)
        if all_real:       # cached device mask: no H2D copy inside the step (keeps it hipGraph-capturable)
            cache = getattr(self, "_all_real_dev", None)
            if cache is None or cache.numel() != B or cache.device != dev:
                cache = self._all_real_dev = torch.ones(B, dtype=torch.bool, device=dev)
            is_real = cache
        else:
            is_real = (isReal if isinstance(isReal, torch.Tensor) else torch.as_tensor(ir_host)).to(dev)
        class_labels = class_labels.to(dev, non_blocking=True)

        if fused_ok and all_real:
            return self._forward_backward_fused(x, class_labels, P, K)

        _, features = self.backbone(x)                                        # :59

        contrastive_loss_query, _, _ = self.contrastive_loss(features, class_labels,
                                                             mask=None if all_real else is_real)  # :62-67
        contrastive_loss_query = contrastive_loss_query * hp.SOLVER.QUERY_CONTRASTIVE_WEIGHT
        if all_real:
            class_labels_real, features_real = class_labels, features
        else:
            ridx = torch.as_tensor(np.nonzero(ir_host)[0], device=dev)
            class_labels_real, features_real = class_labels.index_select(0, ridx), features.index_select(0, ridx)
        center_loss = hp.SOLVER.CENTER_LOSS_WEIGHT * self.center_loss(features_real, class_labels_real)  # :71-73
        bn_features = self.bn(features_real)
        cls_score = self.fc_query(bn_features)
        xent_query = self.xent(cls_score, class_labels_real) * hp.SOLVER.QUERY_XENT_WEIGHT          # :74-77

        # ---- leave-one-out centroids (:79-104) and the K centroid rounds (:112-148)
        centroids_emb, _valid = ops.LooCentroids.apply(features, is_real, P, K)
        ir2 = ir_host.reshape(P, K)
        labels_pk = class_labels.view(P, K)
        feats_pk = features.view(P, K, -1)
        losses, aps, ans, norms = [], [], [], []
        for i in range(K):
            others = ir2.copy(); others[:, i] = False
            valid_p = ir2[:, i] & (others.sum(1) > 0)                       # centroid exists for pid p
            if int(valid_p.sum()) <= 1:                                        # :113-114
                continue
            q_p = ir2[:, i]                                                    # real i-th instances = queries
            if not np.array_equal(q_p, valid_p):
                raise RuntimeError("query/centroid count mismatch in a centroid round (the reference fails in "
                                   "labels.expand at losses/triplet_loss.py:88)")
            if q_p.all():
                query_feat, cur_labels, cur_cent = feats_pk[:, i], labels_pk[:, i], centroids_emb[i]
            else:
                sel = torch.as_tensor(np.nonzero(q_p)[0], device=dev)
                query_feat = feats_pk[:, i].index_select(0, sel)
                cur_labels = labels_pk[:, i].index_select(0, sel)
                cur_cent = centroids_emb[i].index_select(0, sel)
            emb = torch.cat((query_feat, cur_cent))
            lab = torch.cat((cur_labels, cur_labels))
            cos = self.contrastive_loss.dist_name == "cosine"            # SOLVER.DISTANCE_FUNC (:129-137 of the loss)
            if cos:
                emb = ops.RowNormalize.apply(emb, 0, 1e-12)
            loss_i, dap, dan, stats = ops.TripletHardMine.apply(emb, lab, None, self.contrastive_loss.margin, cos)
            losses.append(loss_i); aps.append(stats[1]); ans.append(stats[2])
            norms.append(torch.linalg.vector_norm(cur_cent.detach(), dim=1).mean())
        contrastive_loss_step = torch.mean(torch.stack(losses)) * hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT
        dist_ap = torch.mean(torch.stack(aps))
        dist_an = torch.mean(torch.stack(ans))
        l2_mean_norm_total = torch.mean(torch.stack(norms))

        total_loss = contrastive_loss_step + center_loss + xent_query + contrastive_loss_query       # :150-152
        self.manual_backward(total_loss, optimizer=opt)

        for name, val in zip(self.losses_names, (xent_query, contrastive_loss_query, center_loss, contrastive_loss_step)):
            self.losses_dict[name].append(val.detach())
        log_data = {"step_dist_ap": dist_ap.detach(), "step_dist_an": dist_an.detach(),
                    "l2_mean_centroid": l2_mean_norm_total.detach()}
        return {"loss": total_loss.detach(), "other": log_data}

    # ------------------------------------------------------------------ hand-scheduled heads (all-real batch)
    def _forward_backward_fused(self, x, class_labels, P, K, real=None):
        """The same arithmetic as the autograd path of forward_backward (train_ctl_model.py:59-152), issued as
        one explicit forward/backward schedule over the C ABI: every backward kernel accumulates into a single
        dfeat buffer / the parameters' .grad, loss weights ride in the kernels' gscale argument, and the K centroid
        rounds share one [K, 2P, D] embedding buffer.  real (uint8 [B] on the device, optional): the isReal mask -- the
        masked schedule of train_ctl_model.py:62-148 (anchors of the query triplet, rows of center / BNNeck / xent, rows and
        validity of the centroid rounds), every count taken on the device."""
        hp = self.hparams
        masked = real is not None
        lib, st = L.lib(), L.stream()
        eng = self.backbone.engine
        _, feat = eng.forward(x.contiguous().float(), True, False)            # :59  [B, D] fp32
        B, D = feat.shape
        dev = feat.device
        labels = class_labels.to(torch.int64).contiguous()
        margin = float(self.contrastive_loss.margin)
        if self.heads_one_call and B <= 256 and D % 8 == 0 and getattr(eng, "saved", None) is not None:
            return self._heads_one_call(eng, feat, labels, P, K, real)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        zbuf = torch.zeros(B * D + K * 2 * P * D, **f32)                       # ONE fill for both accumulation buffers
        dfeat = zbuf[:B * D].view(B, D)
        scal = torch.empty(4 * (K + 1) + 2, **f32)                             # every scalar of the step in one buffer:
        out4 = scal[:4 * (K + 1)].view(K + 1, 4)                               # row 0: query triplet, 1..K: rounds
        lc, lx = scal[4 * (K + 1):4 * (K + 1) + 1], scal[4 * (K + 1) + 1:]

        def grad_of(p):
            if not p.requires_grad:
                return None
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            return p.grad

        def triplet(emb, lab, nb, N, o4, gscale, demb, anchor_mask=None, rows=None, gscale_dev=None):
            """nb stacked problems [nb, N, D] in one launch per kernel (mining, loss, backward).  anchor_mask: the reference's
            `mask` argument (anchors dropped after mining); rows: uint8 [nb, N] of rows that exist at all (centroid rounds of
            a batch with fakes; problems with < 2 identities are skipped by the kernel)."""
            dap, dan, coef = torch.empty(nb * N, **f32), torch.empty(nb * N, **f32), torch.empty(nb * N, **f32)
            pi, ni = torch.empty(nb * N, **i32), torch.empty(nb * N, **i32)
            if rows is None:
                L.check(lib.creid_triplet_fwd_batched(L.ptr(emb), L.ptr(lab), L.ptr(anchor_mask), nb, N, D, margin, L.ptr(dap),
                                                      L.ptr(dan), L.ptr(pi), L.ptr(ni), L.ptr(coef), L.ptr(o4), None, st),
                        "creid_triplet_fwd_batched")
            else:
                L.check(lib.creid_triplet_fwd_batched_rows(L.ptr(emb), L.ptr(lab), L.ptr(rows), nb, N, D, margin, 4, L.ptr(dap),
                                                           L.ptr(dan), L.ptr(pi), L.ptr(ni), L.ptr(coef), L.ptr(o4), st),
                        "creid_triplet_fwd_batched_rows")
                L.check(lib.creid_ctl_round_scale(L.ptr(o4), nb, L.ptr(gscale_dev), st), "creid_ctl_round_scale")
            L.check(lib.creid_triplet_bwd_batched(L.ptr(emb), nb, N, D, L.ptr(dap), L.ptr(dan), L.ptr(pi), L.ptr(ni),
                                                  L.ptr(coef), L.ptr(gscale_dev), float(gscale), L.ptr(demb), st),
                    "creid_triplet_bwd_batched")
            return dap, dan, pi, ni, coef                                       # keep alive until the caller returns

        keep = [triplet(feat, labels, 1, B, out4[0], hp.SOLVER.QUERY_CONTRASTIVE_WEIGHT, dfeat, anchor_mask=real)]   # :62-67

        centers = self.center_loss.centers                                     # :71-73
        C_cent = centers.shape[0]
        row_c = torch.empty(B, **f32)
        if masked:
            L.check(lib.creid_center_loss_fwd_masked(L.ptr(feat), L.ptr(labels), L.ptr(centers), L.ptr(real), B, C_cent, D,
                                                     L.ptr(row_c), L.ptr(lc), st), "creid_center_loss_fwd_masked")
            L.check(lib.creid_center_loss_bwd_masked(L.ptr(feat), L.ptr(labels), L.ptr(centers), L.ptr(row_c), L.ptr(real), B, D,
                                                     None, float(hp.SOLVER.CENTER_LOSS_WEIGHT), L.ptr(dfeat),
                                                     L.ptr(grad_of(centers)), st), "creid_center_loss_bwd_masked")
        else:
            L.check(lib.creid_center_loss_fwd(L.ptr(feat), L.ptr(labels), L.ptr(centers), B, C_cent, D, L.ptr(row_c),
                                              L.ptr(lc), st), "creid_center_loss_fwd")
            L.check(lib.creid_center_loss_bwd(L.ptr(feat), L.ptr(labels), L.ptr(centers), L.ptr(row_c), B, D, None,
                                              float(hp.SOLVER.CENTER_LOSS_WEIGHT), L.ptr(dfeat), L.ptr(grad_of(centers)), st),
                    "creid_center_loss_bwd")

        bn, W = self.bn, self.fc_query.weight                                  # :74-77 BNNeck -> classifier -> xent
        bn.num_batches_tracked += 1
        bnf, sm, si = torch.empty((B, D), **f32), torch.empty(D, **f32), torch.empty(D, **f32)
        if masked:
            L.check(lib.creid_bn1d_fwd_masked(L.ptr(feat), L.ptr(real), B, D, L.ptr(bn.weight), L.ptr(bn.bias),
                                              L.ptr(bn.running_mean), L.ptr(bn.running_var), float(bn.momentum), float(bn.eps),
                                              L.ptr(bnf), L.ptr(sm), L.ptr(si), st), "creid_bn1d_fwd_masked")
        else:
            L.check(lib.creid_bn1d_fwd(L.ptr(feat), B, D, L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
                                       L.ptr(bn.running_var), 1, float(bn.momentum), float(bn.eps), L.ptr(bnf), L.ptr(sm),
                                       L.ptr(si), st), "creid_bn1d_fwd")
        C_cls = W.shape[0]
        det = ops._DETERMINISTIC
        logits = ops.gemm_f32(bnf, D, 1, W, 1, D, B, C_cls, D, split_k=1 if det else 32)          # 64-deep K slices
        row_x, dlogits = torch.empty(B, **f32), torch.empty((B, C_cls), **f32)
        if masked:             # padded rows: bnf = 0 -> logits 0, row loss 0, dlogits 0 (no contribution to dW / dbnf)
            L.check(lib.creid_xent_ls_masked(L.ptr(logits), L.ptr(labels), L.ptr(real), B, C_cls, float(self.xent.epsilon),
                                             float(hp.SOLVER.QUERY_XENT_WEIGHT), L.ptr(row_x), L.ptr(lx), L.ptr(dlogits), st),
                    "creid_xent_ls_masked")
        else:
            L.check(lib.creid_xent_ls(L.ptr(logits), L.ptr(labels), B, C_cls, float(self.xent.epsilon),
                                      float(hp.SOLVER.QUERY_XENT_WEIGHT), L.ptr(row_x), L.ptr(lx), L.ptr(dlogits), st),
                    "creid_xent_ls")
        dbnf = ops.gemm_f32(dlogits, C_cls, 1, W, D, 1, B, D, C_cls, split_k=1 if det else 12)      # dlogits @ W
        if W.requires_grad:
            ops.gemm_f32(dlogits, 1, C_cls, bnf, D, 1, C_cls, D, B, out=grad_of(W), beta=1.0)     # += dlogits^T @ bnf
        if masked:
            L.check(lib.creid_bn1d_bwd_masked(L.ptr(feat), L.ptr(dbnf), L.ptr(real), B, D, L.ptr(bn.weight), L.ptr(sm), L.ptr(si),
                                              L.ptr(dfeat), L.ptr(grad_of(bn.weight)), L.ptr(grad_of(bn.bias)), st),
                    "creid_bn1d_bwd_masked")
        else:
            L.check(lib.creid_bn1d_bwd(L.ptr(feat), L.ptr(dbnf), B, D, L.ptr(bn.weight), L.ptr(sm), L.ptr(si), L.ptr(dfeat),
                                       L.ptr(grad_of(bn.weight)), L.ptr(grad_of(bn.bias)), st), "creid_bn1d_bwd")

        # ---- leave-one-out centroids and the K centroid rounds (:79-148)
        cent = torch.empty((K, P, D), **f32)
        valid = torch.empty((K, P), **i32)
        emb = torch.empty((K, 2 * P, D), **f32)                                # round i: P queries, then P centroids
        lab = torch.empty((K, 2 * P), dtype=torch.int64, device=dev)
        cnorm = torch.empty(K * P, **f32)
        demb = zbuf[B * D:].view(K, 2 * P, D)
        if masked:
            rows = torch.empty((K, 2 * P), dtype=torch.uint8, device=dev)
            inv_rounds = torch.empty(1, **f32)
            # (the kernel also counts real instances without a real partner into a persistent device counter: the reference
            # raises on such a batch, this sync-free step cannot -- training_epoch_end / check_lonely_identities() does)
            lonely = self._lonely_counter(dev)
            L.check(lib.creid_loo_emb_fwd_rows_lonely(L.ptr(feat), L.ptr(real), L.ptr(labels), P, K, D, L.ptr(cent), L.ptr(valid),
                                                      L.ptr(emb), L.ptr(lab), L.ptr(cnorm), L.ptr(rows), L.ptr(lonely), st),
                    "creid_loo_emb_fwd_rows_lonely")
            # the K rounds, valid ones only (>= 2 identities kept, :113); their mean is taken on the device (inv_rounds)
            keep.append(triplet(emb, lab, K, 2 * P, out4[1:], hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT, demb, rows=rows,
                                gscale_dev=inv_rounds))
        else:
            real = self._all_real_u8(B, dev)
            L.check(lib.creid_loo_emb_fwd(L.ptr(feat), L.ptr(real), L.ptr(labels), P, K, D, L.ptr(cent), L.ptr(valid), L.ptr(emb),
                                          L.ptr(lab), L.ptr(cnorm), st), "creid_loo_emb_fwd")
            g_round = hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT / K
            keep.append(triplet(emb, lab, K, 2 * P, out4[1:], g_round, demb))      # the K rounds: one launch per kernel
        L.check(lib.creid_loo_emb_bwd(L.ptr(demb), L.ptr(real), P, K, D, L.ptr(dfeat), st), "creid_loo_emb_bwd")

        eng.backward(dfeat)                                                    # manual_backward (:152)

        # weighted terms with ONE multiply: w is a cached constant vector aligned with `scal`
        # (query triplet loss, round losses / K, center loss, xent) -- the logged values are views of the product
        wv = self._loss_weight_vector(K, dev, full_round_weight=masked)
        n = scal.numel()
        stats = torch.empty(n + 7, **f32)                                      # terms[n], total, step, rounds[4], l2
        if masked:
            L.check(lib.creid_ctl_step_stats_rows(L.ptr(scal), L.ptr(wv), n, K, P, L.ptr(cnorm), L.ptr(rows), L.ptr(stats), st),
                    "creid_ctl_step_stats_rows")
        else:
            L.check(lib.creid_ctl_step_stats(L.ptr(scal), L.ptr(wv), n, K, L.ptr(cnorm), K * P, L.ptr(stats), st),
                    "creid_ctl_step_stats")
        terms = stats[:n]
        contrastive_loss_query = terms[0]
        center_loss, xent_query = terms[4 * (K + 1)], terms[4 * (K + 1) + 1]
        total_loss, contrastive_loss_step = stats[n], stats[n + 1]              # :150 (unweighted slots have weight 0)
        rounds = stats[n + 2:n + 6]                                            # {loss, mean ap, mean an, n}
        l2_mean = stats[n + 6]
        for name, val in zip(self.losses_names, (xent_query, contrastive_loss_query, center_loss, contrastive_loss_step)):
            self.losses_dict[name].append(val)
        log_data = {"step_dist_ap": rounds[1], "step_dist_an": rounds[2], "l2_mean_centroid": l2_mean}
        return {"loss": total_loss, "other": log_data}

    def _heads_one_call(self, eng, feat, labels, P, K, real):
        """train_ctl_model.py:59-152 after the backbone forward, through creid_ctl_heads_fused (six launches; the arithmetic and
        the accumulation order are those of the separate calls in _forward_backward_fused: bit-identical results)."""
        import ctypes as C
        hp = self.hparams
        lib = L.lib()
        B, D = feat.shape
        dev = feat.device
        masked = real is not None
        bn, W, centers = self.bn, self.fc_query.weight, self.center_loss.centers
        C_cls = W.shape[0]
        h, w = eng.saved["final"]

        def grad_of(p):
            if not p.requires_grad:
                return None
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            return p.grad

        n = 4 * (K + 1) + 2
        nbytes = lib.creid_ctl_heads_workspace_bytes(B, P, K, D, C_cls)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stats = torch.empty(n + 7, dtype=torch.float32, device=dev)
        g = torch.empty((B * h * w, D), dtype=eng.dtype, device=dev)
        lonely = None
        if masked:
            lonely = self._lonely_counter(dev)
        a = L.CtlHeads()
        a.B, a.P, a.K, a.D, a.num_classes, a.num_centers, a.HW = B, P, K, D, C_cls, centers.shape[0], h * w
        a.g_dtype, a.masked = eng.dt, 1 if masked else 0
        det = ops._DETERMINISTIC
        a.split_logits, a.split_dbnf = (1, 1) if det else (32, 12)
        a.margin, a.xent_eps = float(self.contrastive_loss.margin), float(self.xent.epsilon)
        a.w_query, a.w_center = float(hp.SOLVER.QUERY_CONTRASTIVE_WEIGHT), float(hp.SOLVER.CENTER_LOSS_WEIGHT)
        a.w_xent, a.w_centroid = float(hp.SOLVER.QUERY_XENT_WEIGHT), float(hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT)
        a.bn_momentum, a.bn_eps = float(bn.momentum), float(bn.eps)
        wv = self._loss_weight_vector(K, dev, full_round_weight=masked)
        scaler = eng.loss_scaler if eng.dtype == torch.float16 else None
        keep = (real if masked else self._all_real_u8(B, dev), grad_of(centers), grad_of(bn.weight), grad_of(bn.bias), grad_of(W))
        for name, t in (("feat", feat), ("labels", labels), ("is_real", keep[0]), ("centers", centers), ("bn_weight", bn.weight),
                        ("bn_bias", bn.bias), ("bn_running_mean", bn.running_mean), ("bn_running_var", bn.running_var),
                        ("fc_weight", W), ("loss_weights", wv), ("amp_state", scaler.state if scaler is not None else None),
                        ("d_centers", keep[1]), ("d_bn_weight", keep[2]), ("d_bn_bias", keep[3]), ("d_fc_weight", keep[4]),
                        ("bn_batches_tracked", bn.num_batches_tracked if bn.num_batches_tracked.is_cuda else None),
                        ("lonely", lonely), ("stats", stats), ("g", g), ("dfeat_out", None), ("workspace", ws)):
            setattr(a, name, None if t is None else t.data_ptr())
        a.workspace_bytes = nbytes
        bnred = eng.first_bn_bwd_operands() if hasattr(eng, "first_bn_bwd_operands") else None
        if bnred is not None:
            a.bn_x, a.bn_mask, a.bn_mean, a.bn_invstd, a.bn_partial = (t.data_ptr() for t in bnred)
        if not bn.num_batches_tracked.is_cuda:
            bn.num_batches_tracked += 1
        L.check(lib.creid_ctl_heads_fused(C.byref(a), L.stream()), "creid_ctl_heads_fused")
        if bnred is not None:
            eng.backward(None, g=g, part3=bnred[4])                           # manual_backward (:152)
        else:
            eng.backward(None, g=g)
        terms = stats[:n]
        contrastive_loss_query = terms[0]
        center_loss, xent_query = terms[4 * (K + 1)], terms[4 * (K + 1) + 1]
        total_loss, contrastive_loss_step = stats[n], stats[n + 1]
        rounds = stats[n + 2:n + 6]
        for name, val in zip(self.losses_names, (xent_query, contrastive_loss_query, center_loss, contrastive_loss_step)):
            self.losses_dict[name].append(val)
        log_data = {"step_dist_ap": rounds[1], "step_dist_an": rounds[2], "l2_mean_centroid": stats[n + 6]}
        return {"loss": total_loss, "other": log_data}

    def _lonely_counter(self, dev):
        """Persistent device counter of real instances without a real partner (see check_lonely_identities).  Created OUTSIDE
        any hipGraph capture: a zero-fill captured with the first masked step would reset the counter on every replay."""
        lonely = getattr(self, "_lonely_dev", None)
        if lonely is None or lonely.device != dev:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the first masked training step must run eagerly (or call model._lonely_counter(device) before "
                                   "capturing): the lonely-identity counter cannot be created inside a hipGraph capture")
            lonely = self._lonely_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        return lonely

    def _loss_weight_vector(self, K, dev, full_round_weight=False):
        """full_round_weight: the round slots carry the whole centroid weight (the masked schedule divides by the number of
        VALID rounds on the device) instead of weight / K."""
        hp = self.hparams
        key = (K, str(dev), hp.SOLVER.QUERY_CONTRASTIVE_WEIGHT, hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT,
               hp.SOLVER.CENTER_LOSS_WEIGHT, hp.SOLVER.QUERY_XENT_WEIGHT, bool(full_round_weight))
        cache = self.__dict__.setdefault("_loss_wv", {})
        if key not in cache:
            w = torch.zeros(4 * (K + 1) + 2)
            w[0] = hp.SOLVER.QUERY_CONTRASTIVE_WEIGHT
            w[4:4 * (K + 1):4] = hp.SOLVER.CENTROID_CONTRASTIVE_WEIGHT / (1 if full_round_weight else K)
            w[4 * (K + 1)] = hp.SOLVER.CENTER_LOSS_WEIGHT
            w[4 * (K + 1) + 1] = hp.SOLVER.QUERY_XENT_WEIGHT
            cache[key] = w.to(dev)
        return cache[key]

    def _all_real_u8(self, B, dev):
        c = getattr(self, "_all_real_u8_dev", None)
        if c is None or c.numel() != B or c.device != dev:
            c = self._all_real_u8_dev = torch.ones(B, dtype=torch.uint8, device=dev)
        return c
