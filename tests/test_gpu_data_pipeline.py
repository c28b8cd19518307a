"""GPU parity of the data-pipeline kernels (csrc/preprocess.hip) against the C oracle, through the C-ABI: bit-exact
(integer resize arithmetic, IEEE unfused normalisation, bf16 nearest-even)."""
import os

import numpy as np
import pytest
import torch

from mxdetection_amd._lib import MxdetError
from mxdetection_amd.datasets import append_flipped, synthetic_roidb
from mxdetection_amd.datasets.loader import DetectionLoader
from mxdetection_amd.process_data import transform as T

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_small.npz"))
MEAN, STD = (123.68, 116.779, 103.939), (58.4, 57.1, 57.4)


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def test_preprocess_golden(hip):
    pre = T.BatchPreprocessor(target_size=0, means=MEAN, stds=STD, swap_rb=True, pad_to=(32, 32))
    frames = [torch.from_numpy(G["dp_im0"]).cuda(), torch.from_numpy(G["dp_im1"]).cuda()]
    # the golden case uses free scales: drive the C-ABI plan by hand
    scales = G["dp_scales"].tolist()
    pre.plan = lambda shapes: (scales, [T.resized_shape(h, w, s) for (h, w), s in zip(shapes, scales)], (32, 32))
    out, info, _ = pre(frames, [False, True])
    assert np.array_equal(_bits(out), G["dp_out"])
    assert info[:, :2].tolist() == [[18, 23], [13, 7]]


@pytest.mark.parametrize("swap", [False, True])
def test_preprocess_coco_sizes_vs_oracle(hip, oracle, swap):
    rng = np.random.default_rng(5)
    shapes = [(480, 640), (427, 640), (333, 1000), (612, 612)]
    ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    flips = [False, True, True, False]
    pre = T.BatchPreprocessor(means=MEAN, stds=STD, swap_rb=swap, pad_to=(800, 1344))
    out, info, scales = pre([torch.from_numpy(i).cuda() for i in ims], flips)
    want = oracle.image_preprocess(ims, scales, flips, 800, 1344, MEAN, STD, swap_rb=swap)
    assert np.array_equal(_bits(out), want)
    assert info[2].tolist() == [444, 1333, np.float32(1.333)]
    # zero padding right of / below every resized frame
    o = out.float().cpu().numpy()
    for n in range(4):
        dh, dw = int(info[n, 0]), int(info[n, 1])
        assert not o[n, :, dh:, :].any() and not o[n, :, :, dw:].any() and o[n, :, :dh, :dw].any()


def test_preprocess_direct_gather_form(hip, oracle):
    """The wide-frame fallback (no LDS row staging) gives the same bits."""
    rng = np.random.default_rng(9)
    ims = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), rng.integers(0, 256, (50, 41, 3), dtype=np.uint8)]
    pre = T.BatchPreprocessor(target_size=96, max_size=160, means=MEAN, stds=STD, swap_rb=True, pad_to=(160, 160))
    frames = [torch.from_numpy(i).cuda() for i in ims]
    a, _, scales = pre(frames, [True, False])
    hip.load().mxdet_debug_preprocess_direct(1)
    try:
        b, _, _ = pre(frames, [True, False])
    finally:
        hip.load().mxdet_debug_preprocess_direct(0)
    want = oracle.image_preprocess(ims, scales, [True, False], 160, 160, MEAN, STD, swap_rb=True)
    assert np.array_equal(_bits(a), want) and np.array_equal(_bits(b), want)


def test_preprocess_downscale_and_single_pixel(hip, oracle):
    rng = np.random.default_rng(6)
    ims = [rng.integers(0, 256, (1500, 2100, 3), dtype=np.uint8), rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)]
    pre = T.BatchPreprocessor(target_size=96, max_size=160, means=MEAN, stds=(1, 1, 1), pad_to=None)
    out, info, scales = pre([torch.from_numpy(i).cuda() for i in ims], [True, False])
    hp, wp = out.shape[2:]
    assert (hp, wp) == (96, 160) and info[1].tolist()[:2] == [96, 96]
    assert np.array_equal(_bits(out), oracle.image_preprocess(ims, scales, [True, False], hp, wp, MEAN, (1, 1, 1), swap_rb=False))


def test_preprocess_errors(hip):
    pre = T.BatchPreprocessor(pad_to=(64, 64))
    with pytest.raises(MxdetError, match="does not fit"):
        pre([torch.zeros((480, 640, 3), dtype=torch.uint8, device="cuda")])
    pre = T.BatchPreprocessor(pad_to=(800, 1344), stds=(1.0, 0.0, 1.0))
    with pytest.raises(MxdetError, match="std"):
        pre([torch.zeros((480, 640, 3), dtype=torch.uint8, device="cuda")])


def test_polygon_masks_golden_and_random(hip, oracle):
    dev = "cuda"
    v, s, f = (torch.from_numpy(G[k]).to(dev) for k in ("pm_verts", "pm_start", "pm_first"))
    m = T.polygon_masks(v, s, f, 1, 3, 16, 24)
    assert np.array_equal(m.cpu().numpy()[0], G["pm_masks"])
    # random star-shaped and self-intersecting polygons, vertices on and off pixel centres, one empty instance
    rng = np.random.default_rng(8)
    H, W, N, Gm = 120, 200, 2, 5
    polys = []
    for n in range(N):
        row = []
        for g in range(Gm):
            if g == 3:
                row.append([])
                continue
            inst = []
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(3, 30))
                p = rng.uniform([-10, -10], [W + 10, H + 10], (k, 2)).astype(np.float32)
                if rng.random() < 0.5:
                    p = np.round(p * 2) / 2                      # vertices exactly on pixel centres / edges
                inst.append(p)
            row.append(inst)
        polys.append(row)
    pv, ps, pf = T.pack_polygons(polys, N, Gm)
    m = T.polygon_masks(torch.from_numpy(pv).to(dev), torch.from_numpy(ps).to(dev), torch.from_numpy(pf).to(dev), N, Gm, H, W)
    want = oracle.polygon_masks(pv, ps, pf, N * Gm, H, W).reshape(N, Gm, H, W)
    assert np.array_equal(m.cpu().numpy(), want)
    assert not want[:, 3].any() and 0.05 < want[:, 0].mean() < 0.95


def test_loader_batches_match_host_assembly(hip, oracle):
    roidb = append_flipped(synthetic_roidb(6, seed=4))
    L = DetectionLoader(roidb, 2, with_masks=True, g_max=16, seed=3, num_workers=2)
    want_batches = L.rank_batches()
    seen = 0
    for k, batch in enumerate(L):
        hb = L.assemble(want_batches[k])
        hp, wp = hb.pad
        torch.cuda.synchronize()
        assert tuple(batch["image"].shape) == (2, 3, hp, wp)
        want = oracle.image_preprocess(hb.frames, hb.scales, hb.flips, hp, wp, T.PIXEL_MEANS, T.PIXEL_STDS, swap_rb=False)
        assert np.array_equal(_bits(batch["image"]), want)
        assert np.array_equal(batch["gt_boxes"].cpu().numpy(), hb.gt)
        assert np.array_equal(batch["im_info"].cpu().numpy(), hb.im_info)
        if k < 2:
            wm = oracle.polygon_masks(*hb.poly, 2 * 16, hp, wp).reshape(2, 16, hp, wp)
            got = batch["gt_masks"].cpu().numpy()
            assert np.array_equal(got, wm)
            # the mask of object g fills most of its box and nothing outside it (ellipse inscribed in the box)
            x1, y1, x2, y2 = hb.gt[0, 0, :4]
            ys, xs = np.nonzero(got[0, 0])
            assert xs.min() >= np.floor(x1) - 1 and xs.max() <= np.ceil(x2) + 1 and ys.min() >= np.floor(y1) - 1
            assert 0.6 < got[0, 0].sum() / ((x2 - x1 + 1) * (y2 - y1 + 1)) < 0.9
        seen += 1
    assert seen == len(L) == want_batches.shape[0]


def test_loader_reader_failure_surfaces(hip):
    roidb = synthetic_roidb(6, seed=4)
    calls = {"n": 0}

    def flaky(e):
        calls["n"] += 1
        if calls["n"] > 4:
            raise IOError("disk gone")
        from mxdetection_amd.datasets.synthetic import synthetic_reader
        return synthetic_reader(e)
    L = DetectionLoader(roidb, 2, reader=flaky, shuffle=False, num_workers=1)
    with pytest.raises(RuntimeError, match="reader thread failed"):
        for _ in L:
            pass
