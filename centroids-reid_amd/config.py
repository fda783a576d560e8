"""Configuration tree with the reference's key names (the drop-in contract).

Mirrors the keys of the reference's yacs tree (config/defaults.py:13-181) as a plain
attribute-dict (yacs is not a dependency).  Only the keys the hot path honours carry
meaning here (SURVEY.md §5 "Config / flags"); the rest are kept so that the reference's
YAML files and ``KEY VALUE`` command-line overrides merge unchanged.
"""
from __future__ import annotations

import copy


class CfgNode(dict):
    """Nested dict with attribute access, `clone`, `merge_from_file`, `merge_from_list`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def _set_path(self, dotted, value):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in node:
                raise KeyError(f"Non-existent config key: {dotted}")
            node = node[p]
        if parts[-1] not in node:
            raise KeyError(f"Non-existent config key: {dotted}")
        node[parts[-1]] = value

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                if k not in self:
                    raise KeyError(f"Non-existent config key: {k}")
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        import ast
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for k, v in zip(opts[0::2], opts[1::2]):
            if isinstance(v, str):
                try:
                    v = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            self._set_path(k, v)

    def freeze(self):
        pass


def _node(**kw):
    n = CfgNode()
    n.update(kw)
    return n


def get_cfg_defaults() -> CfgNode:
    """Defaults equal to the reference's (config/defaults.py)."""
    c = CfgNode()
    c.MODEL = _node(NAME="resnet50", BACKBONE_EMB_SIZE=2048, LAST_STRIDE=1, PRETRAINED=True,
                    PRETRAIN_PATH="", USE_CENTROIDS=False, KEEP_CAMID_CENTROIDS=True,
                    RESUME_TRAINING=False)
    c.INPUT = _node(SIZE_TRAIN=[256, 128], SIZE_TEST=[256, 128], PROB=0.5, RE_PROB=0.5,
                    PIXEL_MEAN=[0.485, 0.456, 0.406], PIXEL_STD=[0.229, 0.224, 0.225], PADDING=10)
    c.DATASETS = _node(NAMES="market1501", ROOT_DIR="/home/data", JSON_TRAIN_PATH="")
    c.DATALOADER = _node(NUM_WORKERS=6, SAMPLER="random_identity", NUM_INSTANCE=4, DROP_LAST=True,
                         USE_RESAMPLING=True)
    c.SOLVER = _node(OPTIMIZER_NAME="Adam", MAX_EPOCHS=120, BASE_LR=1e-4, MOMENTUM=0.9, MARGIN=0.5,
                     DISTANCE_FUNC="euclidean", CLUSTER_MARGIN=0.3, CENTER_LR=0.5,
                     CENTER_LOSS_WEIGHT=0.0005, WEIGHT_DECAY=0.0005, WEIGHT_DECAY_BIAS=0.0005,
                     LR_SCHEDULER_NAME="multistep_lr", GAMMA=0.1, LR_STEPS=(40, 70),
                     USE_WARMUP_LR=True, WARMUP_EPOCHS=10, MONITOR_METRIC_NAME="mAP",
                     MONITOR_METRIC_MODE="max", CHECKPOINT_PERIOD=50, EVAL_PERIOD=5,
                     IMS_PER_BATCH=64, DIST_BACKEND="ddp", QUERY_XENT_WEIGHT=1.0,
                     QUERY_CONTRASTIVE_WEIGHT=1.0, CENTROID_CONTRASTIVE_WEIGHT=1.0,
                     USE_AUTOMATIC_OPTIM=False)
    c.TEST = _node(IMS_PER_BATCH=128, WEIGHT="", FEAT_NORM=True, ONLY_TEST=False, VISUALIZE="no",
                   VISUALIZE_TOPK=10, VISUALIZE_MAX_NUMBER=1000000)
    c.GPU_IDS = [0]
    c.LOG_DIR = "logs"
    c.USE_MIXED_PRECISION = True
    c.OUTPUT_DIR = ""
    c.REPRODUCIBLE = False
    c.REPRODUCIBLE_NUM_RUNS = 3
    c.REPRODUCIBLE_SEED = 0
    return c


cfg = get_cfg_defaults()
