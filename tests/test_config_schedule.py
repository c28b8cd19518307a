"""Config system + learning-rate schedule."""
import glob
import os

import pytest

from mxdetection_amd.utils import Config, WarmupMultiFactorScheduler, default_config, epoch_steps, load_config, scaled_lr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attribute_access_and_defaults():
    cfg = default_config()
    assert cfg.TRAIN.lr == 0.02 and cfg["TRAIN"]["batch_images"] == 2 and cfg.network.type == "faster_rcnn"
    cfg.TRAIN.lr = 0.01
    cfg.extra = {"a": {"b": 1}}
    assert cfg.TRAIN.lr == 0.01 and isinstance(cfg.extra.a, Config) and cfg.extra.a.b == 1
    assert default_config().TRAIN.lr == 0.02                       # defaults are not shared
    assert cfg.to_dict()["extra"] == {"a": {"b": 1}}
    with pytest.raises(AttributeError):
        cfg.nope


def test_shipped_configs_load():
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yaml")))
    assert len(files) >= 4
    types = {load_config(f).network.type for f in files}
    assert types == {"faster_rcnn", "mask_rcnn", "retinanet"}
    r = load_config(os.path.join(ROOT, "configs", "retinanet_r101_fpn.yaml"))
    assert r.network.backbone_depth == 101 and r.TRAIN.lr == 0.01 and r.TRAIN.momentum == 0.9   # file over defaults


def test_overrides_and_unknown_keys(tmp_path):
    cfg = load_config(None, ["TRAIN.lr=0.005", "TRAIN.lr_step=[3, 5]", "dataset.type=coco", "TRAIN.warmup=false"])
    assert cfg.TRAIN.lr == 0.005 and cfg.TRAIN.lr_step == [3, 5] and cfg.dataset.type == "coco" and cfg.TRAIN.warmup is False
    with pytest.raises(KeyError, match="TRAIN.learning_rate"):
        load_config(None, ["TRAIN.learning_rate=0.1"])
    with pytest.raises(ValueError):
        load_config(None, ["TRAIN.lr"])
    p = tmp_path / "bad.yaml"
    p.write_text("TRAIN: {lr: 0.1}\nnetwrk: {type: x}\n")
    with pytest.raises(KeyError, match="netwrk"):
        load_config(str(p))
    p.write_text("TRAIN: 3\n")
    with pytest.raises(TypeError):
        load_config(str(p))


def test_warmup_multifactor_schedule():
    s = WarmupMultiFactorScheduler(0.02, steps=[100, 150], factor=0.1, warmup_steps=50, warmup_lr=0.02 / 3)
    assert s(0) == pytest.approx(0.02 / 3) and s(25) == pytest.approx(0.02 / 3 + (0.02 - 0.02 / 3) * 0.5)
    assert s(50) == 0.02 and s(99) == 0.02
    assert s(100) == pytest.approx(0.002) and s(149) == pytest.approx(0.002) and s(150) == pytest.approx(0.0002)
    assert s(10 ** 6) == pytest.approx(0.0002)
    lrs = [s(i) for i in range(50)]
    assert all(b > a for a, b in zip(lrs, lrs[1:]))                   # monotone ramp
    c = WarmupMultiFactorScheduler(0.02, warmup_steps=10, warmup_lr=0.001, warmup_mode="constant")
    assert c(0) == c(9) == 0.001 and c(10) == 0.02
    assert WarmupMultiFactorScheduler(0.1)(0) == 0.1
    with pytest.raises(ValueError):
        WarmupMultiFactorScheduler(0.1, steps=[5, 5])
    with pytest.raises(ValueError):
        WarmupMultiFactorScheduler(0.1, warmup_mode="cosine")


def test_lr_scaling_and_epoch_steps():
    assert scaled_lr(0.02, 16) == 0.02 and scaled_lr(0.02, 2) == pytest.approx(0.0025)
    assert epoch_steps([8, 11], 7330) == [58640, 80630]
    assert epoch_steps([8, 11], 100, begin_epoch=9) == [200]            # resumed past the first boundary
