"""Experiment configuration (/root/reference/README.md:13 `configs`, README.md:49-52
easydict). An attribute dictionary with the lineage's rule that an experiment file may only set keys that exist in the
defaults (typos fail loudly). Files are YAML, read with yaml.safe_load -- the lineage executes python config modules;
nothing is executed here."""
import copy

import yaml


class Config(dict):
    """dict with attribute access, recursively (easydict's behaviour for the subset used here)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else copy.deepcopy(v)) for k, v in self.items()}


DEFAULTS = {
    "network": {
        "type": "faster_rcnn",            # faster_rcnn | mask_rcnn | retinanet
        "backbone_depth": 50,
        "num_classes": 81,                # incl. background for the two-stage models; retinanet uses num_classes - 1
        "pretrained": "",                 # MXNet .params with ImageNet ResNet weights (utils.params_io)
        "seed": 7,
    },
    "dataset": {
        "type": "synthetic",              # synthetic | coco | voc (ann_file = Annotations directory)
        "ann_file": "", "image_dir": "",
        "num_images": 64,                 # synthetic only
        "target_size": 800, "max_size": 1333,
        "pixel_means": [123.68, 116.779, 103.939], "pixel_stds": [1.0, 1.0, 1.0], "swap_rb": False,
        "fixed_shape": [800, 1344],       # one batch shape (hipGraph replay); [] = per-batch shapes (eager launches)
        "max_gt": 100,
    },
    "TRAIN": {
        "batch_images": 2,                # per GPU
        "lr": 0.02, "lr_reference_batch": 16, "lr_step": [8, 11], "lr_factor": 0.1,
        "warmup": True, "warmup_lr": 0.00667, "warmup_step": 500, "warmup_mode": "linear",
        "momentum": 0.9, "wd": 0.0001,
        "begin_epoch": 0, "end_epoch": 12,
        "flip": True, "shuffle": True, "aspect_grouping": True, "seed": 0,
        "rpn_pre_nms_top_n": 2000, "rpn_post_nms_top_n": 2000, "batch_rois": 512,
        "graph": True,                    # hipGraph replay of the step
        "checkpoint_prefix": "", "checkpoint_period": 1, "resume": "",
        "log_period": 20, "max_iters": 0,
    },
    "TEST": {
        "batch_images": 2, "score_thresh": 0.05, "nms": 0.5, "max_per_image": 100,
        "bbox_means": [0.0, 0.0, 0.0, 0.0], "bbox_stds": [0.1, 0.1, 0.2, 0.2],
    },
}


def default_config():
    return Config(copy.deepcopy(DEFAULTS))


def update_config(cfg, d, _path=""):
    """Merge d into cfg; every key must already exist (lineage `update_config`)."""
    for k, v in d.items():
        if k not in cfg:
            raise KeyError("unknown config key %s%s" % (_path, k))
        if isinstance(cfg[k], Config):
            if not isinstance(v, dict):
                raise TypeError("config key %s%s is a section" % (_path, k))
            update_config(cfg[k], v, _path + k + ".")
        else:
            cfg[k] = v
    return cfg


def load_config(path=None, overrides=()):
    """defaults <- YAML file <- "SECTION.key=value" overrides (values parsed as YAML scalars/lists)."""
    cfg = default_config()
    if path:
        with open(path, "r") as f:
            update_config(cfg, yaml.safe_load(f) or {})
    for o in overrides:
        key, _, val = o.partition("=")
        if not _:
            raise ValueError("override %r is not KEY=VALUE" % (o,))
        node = {}
        cur = node
        parts = key.split(".")
        for p in parts[:-1]:
            cur[p] = {}
            cur = cur[p]
        cur[parts[-1]] = yaml.safe_load(val)
        update_config(cfg, node)
    return cfg
