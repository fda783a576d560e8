"""Import the read-only upstream reference (/root/reference) in THIS container.

The reference depends on packages that are absent here (pytorch_lightning,
yacs, cv2, torchvision, mlflow).  This module installs minimal ``sys.modules``
stand-ins so the reference's torch-only arithmetic can be imported and run on
CPU to generate golden vectors (tools/gen_golden.py).  It is build-time tooling:
nothing under tests/, bench.py or the product package imports it, and the
reference never travels to the GPU box.
"""
import os
import sys
import types

import torch

REF = os.environ.get("CREID_REFERENCE", "/root/reference")


class AttrDict(dict):
    """dict with attribute access (stands in for yacs CfgNode / PL AttributeDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def merge_from_file(self, *_a, **_k):
        pass

    def merge_from_list(self, *_a, **_k):
        pass

    def freeze(self):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_creid_stub", False):
        return

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    class Callback:
        pass

    def rank_zero_only(fn):
        return fn

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, Trainer=object,
              LightningDataModule=object, _creid_stub=True)
    _mod("pytorch_lightning.utilities", AttributeDict=AttrDict, rank_zero_only=rank_zero_only)
    _mod("pytorch_lightning.utilities.seed", seed_everything=lambda *a, **k: None)
    _mod("pytorch_lightning.callbacks", ModelCheckpoint=object, Callback=Callback)
    _mod("pytorch_lightning.callbacks.base", Callback=Callback)
    _mod("pytorch_lightning.loggers", MLFlowLogger=object, TensorBoardLogger=object)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]
    _mod("cv2")
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.transforms = _mod("torchvision.transforms")
    tv.datasets = _mod("torchvision.datasets", ImageFolder=type("ImageFolder", (), {}))
    _mod("torchvision.datasets.folder", default_loader=None, IMG_EXTENSIONS=(), is_image_file=lambda p: True)
    _mod("yacs")
    _mod("yacs.config", CfgNode=AttrDict)
    _mod("mlflow")
    # the reference's loss modules default to use_gpu=True and call .cuda()
    torch.Tensor.cuda = lambda self, *a, **k: self


def ref_path_first():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # the pip `datasets` (HF) package would shadow the reference's datasets/ dir
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]


def load():
    """Returns a namespace with the reference modules on the hot path."""
    install_stubs()
    ref_path_first()
    import importlib
    ns = types.SimpleNamespace()
    ns.triplet_loss = importlib.import_module("losses.triplet_loss")
    ns.center_loss = importlib.import_module("losses.center_loss")
    ns.eval_reid = importlib.import_module("utils.eval_reid")
    ns.resnet = importlib.import_module("modelling.backbones.resnet")
    ns.resnet_ibn_a = importlib.import_module("modelling.backbones.resnet_ibn_a")
    ns.config = importlib.import_module("config")
    ns.reid_metric = importlib.import_module("utils.reid_metric")
    ns.baseline = importlib.import_module("modelling.baseline")
    ns.bases = importlib.import_module("modelling.bases")
    ns.train_ctl_model = importlib.import_module("train_ctl_model")
    ns.solver = importlib.import_module("solver.build")
    return ns


if __name__ == "__main__":
    ns = load()
    print("reference imported:", [k for k in vars(ns)])
