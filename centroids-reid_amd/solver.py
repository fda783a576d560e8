"""Optimisers / scheduler of the reference (solver/build.py:9-63) on fused HIP steps.

build_optimizer keeps the reference's grouping rule (everything whose name contains "center" goes to a
plain SGD(lr=CENTER_LR); the rest to Adam(lr=BASE_LR, weight_decay=WEIGHT_DECAY), parameters with
requires_grad=False skipped, extra "names" key in the param groups).  The Adam group lives in ONE flat
fp32 buffer (parameters and gradients are views into it), so a step is a single kernel and the same
buffer is what the data-parallel all-reduce sends over RCCL.
"""
from __future__ import annotations

import torch

from . import _lib as L


def flat_offsets(params):
    """Start offset of every tensor in the flat buffer (each starts 16-byte aligned) and the padded total."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    return offs, off


def flatten_(params, device, grad_tail=0):
    """Re-home `params` (list of nn.Parameter) as views of one flat fp32 buffer; returns (flat, gflat).  `grad_tail` extra
    elements behind the gradients: room for the gradient of parameters another optimiser owns (the centers), so that ONE
    collective carries both."""
    offs, n_pad = flat_offsets(params)
    flat = torch.zeros(n_pad, dtype=torch.float32, device=device)
    gflat = torch.zeros(n_pad + grad_tail, dtype=torch.float32, device=device)
    for p, off in zip(params, offs):
        k = p.numel()
        flat[off:off + k].copy_(p.data.reshape(-1))
        p.data = flat[off:off + k].view(p.shape)
        p.grad = gflat[off:off + k].view(p.shape)
    return flat, gflat


class LossScaler:
    """Dynamic loss scale of f16 mixed-precision training (the reference's `precision=16`, utils/misc.py:111 = native AMP with
    torch.cuda.amp.GradScaler: init 2**16, x 0.5 on a non-finite gradient, x 2 after 2000 clean steps) with ALL of its state on
    the device: {scale, 1 / scale} and {found_inf, clean steps}.  Nothing here synchronises with the host, so a captured
    hipGraph of the training step replays correctly through overflow steps (the skipped update, the halved scale and Adam's
    un-advanced step counter are all decided by the kernels: csrc/heads.hip amp_*).

    Where it acts: `scale_(dfeat)` multiplies the heads' gradient as it ENTERS the f16 backbone backward (the heads run in
    fp32 and need no scaling; every f16 gradient tensor and every backbone parameter gradient downstream is then scaled);
    FusedAdam.step unscales the backbone prefix of its flat gradient buffer in place, checks it for inf / nan and skips itself
    when one is found; CenterSGD.step skips with it; `update()` applies GradScaler's growth / backoff rule and clears the flag."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 1.0 / init_scale], dtype=torch.float32, device=device)
        self.flags = torch.zeros(3, dtype=torch.int32, device=device)     # {found_inf, clean steps in a row, steps skipped in total}
        self._ones = torch.ones(2, dtype=torch.float32, device=device)   # 'scale 1' for the check-only pass
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)

    def scale_(self, x):
        y = torch.empty_like(x)
        L.check(L.lib().creid_amp_scale(L.ptr(x), x.numel(), L.ptr(self.state), L.ptr(y), L.stream()), "creid_amp_scale")
        return y

    def unscale_check_(self, g):
        L.check(L.lib().creid_amp_unscale_check(L.ptr(g), g.numel(), L.ptr(self.state), L.ptr(self.flags), L.stream()),
                "creid_amp_unscale_check")

    def check_(self, g):
        """Non-finite check WITHOUT unscaling (gradients that never saw the loss scale: BNNeck, classifier, centers).
        GradScaler.step looks at every parameter of the optimiser; so does this, into the same flag."""
        if g.numel():
            L.check(L.lib().creid_amp_unscale_check(L.ptr(g), g.numel(), L.ptr(self._ones), L.ptr(self.flags), L.stream()),
                    "creid_amp_unscale_check")

    @property
    def skipped_steps(self):            # (host read-back: diagnostics / logging)
        return int(self.flags[2].item())

    def update(self):
        L.check(L.lib().creid_amp_update(L.ptr(self.state), L.ptr(self.flags), self.growth_factor, self.backoff_factor,
                                         self.growth_interval, L.stream()), "creid_amp_update")

    def get_scale(self):            # (host read-back: diagnostics / tests only)
        return float(self.state[0].item())

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self.flags[1].item()),
                "_skipped_steps": int(self.flags[2].item())}

    def load_state_dict(self, sd):
        s = float(sd["scale"])
        self.state.copy_(torch.tensor([s, 1.0 / s], dtype=torch.float32))
        self.flags.copy_(torch.tensor([0, int(sd.get("_growth_tracker", 0)), int(sd.get("_skipped_steps", 0))], dtype=torch.int32))
        self.growth_factor, self.backoff_factor = float(sd.get("growth_factor", 2.0)), float(sd.get("backoff_factor", 0.5))
        self.growth_interval = int(sd.get("growth_interval", 2000))


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay in the gradient) -- one kernel over the flat buffer."""

    def __init__(self, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_tail=0):
        super().__init__(param_groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        params = [p for g in self.param_groups for p in g["params"]]
        assert len(self.param_groups) == 1
        self.flat, self.gflat = flatten_(params, params[0].device, grad_tail)
        self.gtail = self.gflat[self.flat.numel():]          # not an Adam gradient: see build_optimizer
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.grad_scale = 1.0
        self.scaler, self.n_scaled = None, 0       # f16 training: LossScaler + length of the loss-scaled prefix of gflat
        # device-resident {lr, step, bc1, bc2s}: keeps a captured hipGraph of the step valid
        self.hyper = torch.zeros(8, dtype=torch.float32, device=self.flat.device)   # [4]: the kernel's ticket (int32 0)
        self._lr_on_device = None
        self._params = params
        self._offsets, _ = flat_offsets(params)
        self._bind_state_views()

    def _bind_state_views(self):
        """torch-compatible per-parameter state for checkpoints: VIEWS of the flat moment buffers, at the same
        (16-byte aligned) offsets as the parameters themselves."""
        for p, off in zip(self._params, self._offsets):
            k = p.numel()
            self.state[p] = dict(step=torch.tensor(0.0), exp_avg=self.exp_avg[off:off + k].view(p.shape),
                                 exp_avg_sq=self.exp_avg_sq[off:off + k].view(p.shape))

    def load_state_dict(self, state_dict):
        """Restore the moments INTO the flat buffers the kernel reads and the step count into the device-resident
        hyper vector (torch's default only swaps self.state[p] for detached copies, which the kernel never sees).
        Accepts this class's own state_dict and a torch.optim.Adam one (same per-parameter keys)."""
        super().load_state_dict(state_dict)
        step = 0.0
        with torch.no_grad():
            for p, off in zip(self._params, self._offsets):
                st = self.state.get(p)
                if not st:
                    continue
                k = p.numel()
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, float(st["step"]))
            self.hyper[1:2].copy_(torch.tensor([step], dtype=torch.float32))     # bc1 / bc2 are recomputed per step
        self._bind_state_views()
        self._lr_on_device = None

    def zero_grad(self, set_to_none: bool = False):
        self.gflat.zero_()

    def attach_scaler(self, scaler, scaled_prefix_names=("backbone.",)):
        """f16 training: gradients of the parameters whose name starts with one of `scaled_prefix_names` (the f16 backbone)
        arrive multiplied by the loss scale; they must form a PREFIX of the flat buffer (they do: the backbone is the first
        module of ModelBase, solver/build.py keeps named_parameters order), which step() unscales and checks in one pass."""
        names = self.param_groups[0].get("names")
        assert names is not None and len(names) == len(self._params), "attach_scaler needs the reference's 'names' key"
        scaled = [n.startswith(tuple(scaled_prefix_names)) for n in names]
        k = sum(scaled)
        assert all(scaled[:k]) and not any(scaled[k:]), "loss-scaled parameters must be a prefix of the flat gradient buffer"
        self.n_scaled = self._offsets[k] if k < len(self._params) else self.flat.numel()
        self.scaler = scaler

    @property
    def step_count(self):
        return int(self.hyper[1].item())

    def state_dict(self):
        n = self.step_count
        for st in self.state.values():
            st["step"] = torch.tensor(float(n))
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        lr = float(g["lr"])
        if lr != self._lr_on_device:          # only when the schedule / warm-up changes it (outside graphs)
            self.hyper[0:1].copy_(torch.tensor([lr], dtype=torch.float32))
            self._lr_on_device = lr
        b1, b2 = g["betas"]
        if self.scaler is not None:
            # (after the data-parallel all-reduce: a non-finite value on any rank has reached every rank's buffer by now, so all
            # ranks skip the same steps without a collective of their own)
            self.scaler.unscale_check_(self.gflat[:self.n_scaled])
            self.scaler.check_(self.gflat[self.n_scaled:])        # heads + the centers' gradient in the tail: checked, not unscaled
            L.check(L.lib().creid_adam_step_dev_amp(L.ptr(self.flat), L.ptr(self.gflat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                                    self.flat.numel(), L.ptr(self.hyper), b1, b2, g["eps"], g["weight_decay"],
                                                    float(self.grad_scale), L.ptr(self.scaler.flags), L.stream()),
                    "creid_adam_step_dev_amp")
            return
        L.check(L.lib().creid_adam_step_dev(L.ptr(self.flat), L.ptr(self.gflat), L.ptr(self.exp_avg),
                                            L.ptr(self.exp_avg_sq), self.flat.numel(), L.ptr(self.hyper), b1, b2,
                                            g["eps"], g["weight_decay"], float(self.grad_scale), L.stream()),
                "creid_adam_step_dev")


class CenterSGD(torch.optim.Optimizer):
    """SGD(lr=CENTER_LR) for CenterLoss.centers with the reference's in-place gradient rescale
    (train_ctl_model.py:157-159) folded into the same kernel: g *= grad_mul; p -= lr * g."""

    def __init__(self, param_groups, lr=0.5):
        super().__init__(param_groups, dict(lr=lr))
        self.grad_mul = 1.0
        self.grad_scale = 1.0          # data-parallel 1 / world when the gradient was SUM-reduced inside the Adam buffer's tail
        self._tail = None
        self.scaler = None             # f16 training: the step is skipped together with Adam's on a non-finite backbone gradient

    @property
    def grad_in_adam_tail(self):
        """True iff EVERY parameter's .grad still is a view of the adopted tail.  Checked at every read (two pointer compares
        per parameter): a foreign wrapper's zero_grad(set_to_none=True), `p.grad = ...` in user code or a re-created Parameter
        rebinds .grad, and the data-parallel sync must then fall back to the explicit all-reduce instead of silently stepping
        on an unreduced gradient with 1 / world applied (parallel.make_grad_sync / make_overlapped_grad_sync read this)."""
        if self._tail is None:
            return False
        lo = self._tail.data_ptr()
        hi = lo + self._tail.numel() * self._tail.element_size()
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    return False
        return True

    def readopt_tail(self):
        """Re-home gradients that were rebound away from the tail (their values are kept).  Outside captured graphs only."""
        if self._tail is None or self.grad_in_adam_tail:
            return
        old = [[None if p.grad is None else p.grad.detach().clone() for p in g["params"]] for g in self.param_groups]
        self.adopt_tail(self._tail)
        for g, og in zip(self.param_groups, old):
            for p, o in zip(g["params"], og):
                p.grad.zero_() if o is None else p.grad.copy_(o)

    def adopt_tail(self, tail):
        """The parameters' gradients become views of `tail` (the room behind FusedAdam's flat gradient buffer): the
        data-parallel all-reduce of that buffer then carries them too -- one collective less per step."""
        off = 0
        for g in self.param_groups:
            for p in g["params"]:
                k = p.numel()
                assert off + k <= tail.numel()
                p.grad = tail[off:off + k].view(p.shape)
                off += (k + 3) // 4 * 4
        self._tail = tail

    def zero_grad(self, set_to_none: bool = False):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if self.scaler is not None:
                    L.check(L.lib().creid_sgd_scaled_step_amp(L.ptr(p.data), L.ptr(p.grad), p.numel(), float(g["lr"]),
                                                              float(self.grad_mul) * float(self.grad_scale),
                                                              L.ptr(self.scaler.flags), L.stream()), "creid_sgd_scaled_step_amp")
                    continue
                L.check(L.lib().creid_sgd_scaled_step(L.ptr(p.data), L.ptr(p.grad), p.numel(), float(g["lr"]),
                                                      float(self.grad_mul) * float(self.grad_scale), L.stream()), "creid_sgd_scaled_step")
        self.grad_mul = 1.0


def build_optimizer(named_parameters, hparams):
    """solver/build.py:9-47."""
    regular, regular_names, center, center_names = [], [], [], []
    for name, parameter in named_parameters:
        if parameter.requires_grad is False:
            print(f"Parameter {name} does not need a Grad. Excluding from the optimizer...")
            continue
        elif "center" in name:
            center.append(parameter); center_names.append(name)
        else:
            regular.append(parameter); regular_names.append(name)
    if hparams.SOLVER.OPTIMIZER_NAME != "Adam":
        raise NotImplementedError(f"No such optimizer {hparams.SOLVER.OPTIMIZER_NAME}")
    # the centers' gradient lives behind the Adam group's flat gradient buffer (never seen by the Adam kernel, whose length is
    # the parameter buffer's): the gradient all-reduce of the last layer group carries it (parallel.py)
    tail = sum((p.numel() + 3) // 4 * 4 for p in center)
    model_optimizer = FusedAdam([{"params": regular, "names": regular_names}], lr=hparams.SOLVER.BASE_LR,
                                weight_decay=hparams.SOLVER.WEIGHT_DECAY, grad_tail=tail)
    optimizer_center = CenterSGD([{"params": center, "names": center_names}], lr=hparams.SOLVER.CENTER_LR)
    if tail:
        optimizer_center.adopt_tail(model_optimizer.gtail)
    return [model_optimizer, optimizer_center]


def build_scheduler(model_optimizer, hparams):
    """solver/build.py:50-63."""
    if hparams.SOLVER.LR_SCHEDULER_NAME == "cosine_annealing":
        return torch.optim.lr_scheduler.CosineAnnealingLR(model_optimizer, hparams.SOLVER.MAX_EPOCHS,
                                                          eta_min=hparams.SOLVER.MIN_LR)
    elif hparams.SOLVER.LR_SCHEDULER_NAME == "multistep_lr":
        return torch.optim.lr_scheduler.MultiStepLR(model_optimizer, milestones=hparams.SOLVER.LR_STEPS,
                                                    gamma=hparams.SOLVER.GAMMA)
    raise NotImplementedError(f"No such scheduler {hparams.SOLVER.LR_SCHEDULER_NAME}")
