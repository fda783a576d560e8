"""Minimal stand-in for pytorch-lightning 1.1.4 (absent from this image), ONLY for tools/debug/pl_stub_check.py: the part of the
LightningModule surface centroids-reid_amd/bases.py touches when `import pytorch_lightning` succeeds -- save_hyperparameters,
optimizers() returning LightningOptimizer WRAPPERS (attributes set on a wrapper do not reach the wrapped optimizer, exactly
the trap `_raw_optimizers` exists for), manual_backward, optimizer_step, current_epoch."""
from types import SimpleNamespace

from torch import nn

__version__ = "1.1.4-stub"


class LightningOptimizer:
    def __init__(self, optimizer):
        # PL-1.1.4 copies the optimizer's __dict__ (minus step) into the wrapper: later attribute writes stay on the wrapper
        self.__dict__ = {k: v for k, v in optimizer.__dict__.items() if k != "step"}
        self._optimizer = optimizer

    def step(self, *args, closure=None, **kwargs):
        if closure is not None:
            closure()
        return self._optimizer.step()

    def zero_grad(self, *a, **k):
        return self._optimizer.zero_grad(*a, **k)


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = SimpleNamespace(current_epoch=0, global_rank=0, local_rank=0, logger=None, train_dataloader=None)
        self._stub_optimizers = None
        self.automatic_optimization = False

    def save_hyperparameters(self, hp):
        self.hparams = hp

    @property
    def current_epoch(self):
        return self.trainer.current_epoch

    def optimizers(self, use_pl_optimizer=True):
        if self._stub_optimizers is None:
            opts, _sched = self.configure_optimizers()
            self._stub_optimizers = list(opts)
        if use_pl_optimizer:
            return [LightningOptimizer(o) for o in self._stub_optimizers]
        return self._stub_optimizers

    def manual_backward(self, loss, optimizer=None, *args, **kwargs):
        loss.backward()

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, optimizer_idx=None, optimizer_closure=None,
                       on_tpu=False, using_native_amp=False, using_lbfgs=False, **kwargs):
        if optimizer_closure is not None:
            optimizer_closure()
        optimizer.step()
