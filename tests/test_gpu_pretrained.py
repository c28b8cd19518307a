"""Pretrained-backbone import (utils/pretrained.py): an MXNet-named ResNet-50 .params blob with live BatchNorm
statistics, folded and loaded, must reproduce a plain fp32 conv -> BN(eval) -> ReLU ResNet-v1b forward (torch CPU).
Tolerance: activations are bf16 between layers (8 mantissa bits) through 16 residual blocks -> relative Frobenius
error below 3e-2 per pyramid level."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mxdetection_amd.utils import load_params, load_pretrained_backbone, resnet_v1_names, save_params

pytestmark = pytest.mark.gpu
EPS = 2e-5


def _random_resnet50_blob(rng):
    blob = {}
    shapes = {"stem": (64, 3, 7, 7)}
    cin = 64
    for si, nb in enumerate((3, 4, 6, 3)):
        planes = 64 << si
        for bi in range(nb):
            n = "layer%d.%d" % (si + 1, bi)
            shapes[n + ".conv1"] = (planes, cin if bi == 0 else planes * 4, 1, 1)
            shapes[n + ".conv2"] = (planes, planes, 3, 3)
            shapes[n + ".conv3"] = (planes * 4, planes, 1, 1)
            if bi == 0:
                shapes[n + ".down"] = (planes * 4, cin, 1, 1)
        cin = planes * 4
    for ours, wname, bn in resnet_v1_names(50):
        co, ci, kh, kw = shapes[ours]
        std = (2.0 / (ci * kh * kw)) ** 0.5 * (0.3 if ours.endswith("conv3") else 1.0)
        blob["arg:" + wname] = (rng.standard_normal((co, ci, kh, kw)) * std).astype(np.float32)
        blob["arg:" + bn + "_gamma"] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        blob["arg:" + bn + "_beta"] = rng.uniform(-0.2, 0.2, co).astype(np.float32)
        blob["aux:" + bn + "_moving_mean"] = rng.uniform(-0.3, 0.3, co).astype(np.float32)
        blob["aux:" + bn + "_moving_var"] = rng.uniform(0.5, 2.0, co).astype(np.float32)
    return blob


def _ref_forward(blob, x):
    def cbr(t, wname, bn, stride=1, pad=0, relu=True):
        t = F.conv2d(t, torch.from_numpy(blob["arg:" + wname]), None, stride, pad)
        t = F.batch_norm(t, torch.from_numpy(blob["aux:" + bn + "_moving_mean"]), torch.from_numpy(blob["aux:" + bn + "_moving_var"]),
                         torch.from_numpy(blob["arg:" + bn + "_gamma"]), torch.from_numpy(blob["arg:" + bn + "_beta"]), False, 0.0, EPS)
        return F.relu(t) if relu else t
    t = cbr(x, "conv0_weight", "bn0", 2, 3)
    t = F.max_pool2d(t, 3, 2, 1)
    outs = []
    for si, nb in enumerate((3, 4, 6, 3)):
        for bi in range(nb):
            u = "stage%d_unit%d" % (si + 1, bi + 1)
            stride = 2 if (bi == 0 and si > 0) else 1
            sc = cbr(t, u + "_sc_weight", u + "_sc_bn", stride, 0, relu=False) if bi == 0 else t
            a = cbr(t, u + "_conv1_weight", u + "_bn1")
            a = cbr(a, u + "_conv2_weight", u + "_bn2", stride, 1)          # v1b: the stride sits on the 3x3
            a = cbr(a, u + "_conv3_weight", u + "_bn3", relu=False)
            t = F.relu(a + sc)
        outs.append(t)
    return outs


def test_pretrained_resnet50_import(hip, tmp_path):
    from mxdetection_amd.models import FasterRCNN
    rng = np.random.default_rng(3)
    blob = _random_resnet50_blob(rng)
    path = str(tmp_path / "resnet-v1-50-0000.params")
    save_params(path, blob)
    assert set(load_params(path)) == set(blob)
    model = FasterRCNN("cuda", depth=50, seed=1)
    assert load_pretrained_backbone(model, path, depth=50) == []
    x = torch.from_numpy(rng.standard_normal((2, 3, 64, 96)).astype(np.float32))
    want = _ref_forward(blob, x)
    model.backbone.plan((2, 3, 64, 96))
    got = model.backbone.forward(x.cuda())
    for lvl, (g, w) in enumerate(zip(got, want)):
        g = g.float().permute(0, 3, 1, 2).cpu()
        assert g.shape == w.shape
        rel = float((g - w).norm() / w.norm())
        assert rel < 3e-2, "C%d relative error %.4f" % (lvl + 2, rel)
    # a file without the shortcut BN statistics is refused, naming what is missing
    del blob["aux:stage2_unit1_sc_bn_moving_var"]
    with pytest.raises(KeyError, match="lacks 1 backbone tensors"):
        load_pretrained_backbone(model, blob, depth=50)
