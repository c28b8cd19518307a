import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def hip():
    """The HIP library through the product's own loader; GPU tests fail loudly if it is missing."""
    import torch
    from mxdetection_amd import _lib
    assert torch.cuda.is_available(), "gpu test without a GPU"
    _lib.load()
    return _lib


def synth_gt(rng, n_img, g_max=16, im_h=800, im_w=1333, g_lo=4, g_hi=16, num_classes=80):
    """SURVEY.md section 8d GT generator: log-uniform sizes in [16,600], uniform position, -1 padding."""
    gt = -np.ones((n_img, g_max, 5), np.float32)
    for n in range(n_img):
        g = int(rng.integers(g_lo, g_hi + 1))
        for k in range(min(g, g_max)):
            w = float(np.exp(rng.uniform(np.log(16), np.log(600))))
            h = float(np.exp(rng.uniform(np.log(16), np.log(600))))
            w, h = min(w, im_w - 1), min(h, im_h - 1)
            x1 = float(rng.uniform(0, im_w - w))
            y1 = float(rng.uniform(0, im_h - h))
            gt[n, k] = [x1, y1, x1 + w - 1, y1 + h - 1, float(rng.integers(1, num_classes + 1))]
    return gt


def synth_boxes(rng, n, im_h=800, im_w=1333):
    w = np.exp(rng.uniform(np.log(8), np.log(500), n))
    h = np.exp(rng.uniform(np.log(8), np.log(500), n))
    x1 = rng.uniform(0, im_w - 1, n)
    y1 = rng.uniform(0, im_h - 1, n)
    x2 = np.minimum(x1 + w, im_w - 1)
    y2 = np.minimum(y1 + h, im_h - 1)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)
