"""Pins include/mxdet_math.h (the only code shared by oracle and product) against independent references."""
import numpy as np


def test_philox_known_answer_vectors(oracle):
    # Random123 kat_vectors, philox4x32-10 (Salmon et al., SC'11)
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for c, k, want in kat:
        got = oracle.philox(c, k)
        assert tuple(int(v) for v in got) == want


def test_expf_logf_against_float64(oracle):
    xs = np.concatenate([np.linspace(-20, 20, 2001), np.array([0.0, 1e-6, -1e-6, 4.135166556742356])]).astype(np.float32)
    got = oracle.expf(xs).astype(np.float64)
    want = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - want) / want) < 4e-7
    ys = np.concatenate([np.exp(np.linspace(-12, 12, 2001)), np.array([1.0, 0.5, 2.0])]).astype(np.float32)
    got = oracle.logf(ys).astype(np.float64)
    want = np.log(ys.astype(np.float64))
    assert np.max(np.abs(got - want)) < 3e-7 * np.maximum(1.0, np.abs(want)).max()
    assert oracle.logf(np.float32(1.0)) == 0.0
    assert oracle.expf(np.float32(0.0)) == 1.0


def test_bf16_round_to_nearest_even(oracle):
    import torch
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 37.0
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    got = oracle.f32_to_bf16_bits(x)
    assert np.array_equal(got, want)
    L = oracle.lib()
    for v in (1.0, 1.00390625, 1.01171875, -3.3895313892515355e38, 65504.0):
        assert L.oracle_f32_to_bf16(v) == int(torch.tensor([v]).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)[0])


def test_decode_encode_roundtrip_and_float64(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        ex = np.sort(rng.uniform(0, 500, 2)).tolist() + np.sort(rng.uniform(0, 500, 2)).tolist()
        ex = np.array([ex[0], ex[2], ex[1] + 8, ex[3] + 8], np.float32)
        gt = np.array([ex[0] + rng.uniform(-5, 5), ex[1] + rng.uniform(-5, 5), ex[2] + rng.uniform(-5, 5) + 3,
                       ex[3] + rng.uniform(-5, 5) + 3], np.float32)
        d = oracle.encode(ex, gt)
        back = oracle.decode_clip(ex, d, 10000.0, 10000.0)
        assert np.allclose(back, np.clip(gt, 0, 9999), atol=2e-3)
        # float64 restatement of the published transform
        ew, eh = ex[2] - ex[0] + 1.0, ex[3] - ex[1] + 1.0
        gw, gh = gt[2] - gt[0] + 1.0, gt[3] - gt[1] + 1.0
        want = [((gt[0] + 0.5 * (gw - 1)) - (ex[0] + 0.5 * (ew - 1))) / ew,
                ((gt[1] + 0.5 * (gh - 1)) - (ex[1] + 0.5 * (eh - 1))) / eh, np.log(gw / ew), np.log(gh / eh)]
        assert np.allclose(d, want, atol=1e-5)


# ---- independent numpy float32 restatements of the box formulas (the specification in DESIGN.md section 3, written
# from the published mx-rcnn / Detectron definitions, NOT from include/mxdet_math.h): IEEE float32 add / sub / mul / div
# round identically everywhere, so with the same operation order the results must agree BIT FOR BIT with the header the
# oracle and the kernels share. (exp / log are the header's own polynomials: those two are pinned against float64 above
# and the decode / encode widths are compared through oracle.expf / oracle.logf here.)
def _f(x):
    return np.float32(x)


def _np_iou(a, b):
    ix1, iy1 = max(a[0], b[0]), max(a[1], b[1])
    ix2, iy2 = min(a[2], b[2]), min(a[3], b[3])
    iw, ih = _f(_f(ix2 - ix1) + _f(1)), _f(_f(iy2 - iy1) + _f(1))
    if iw <= 0 or ih <= 0:
        return _f(0)
    inter = _f(iw * ih)
    aa = _f(_f(_f(a[2] - a[0]) + _f(1)) * _f(_f(a[3] - a[1]) + _f(1)))
    ab = _f(_f(_f(b[2] - b[0]) + _f(1)) * _f(_f(b[3] - b[1]) + _f(1)))
    return _f(inter / _f(_f(aa + ab) - inter))


def test_iou_fpn_level_smooth_l1_independent_restatement(oracle):
    rng = np.random.default_rng(123)
    a = rng.uniform(0, 600, (40, 2)).astype(np.float32)
    a = np.concatenate([a, a + rng.uniform(0, 300, (40, 2)).astype(np.float32)], 1)
    b = rng.uniform(0, 600, (30, 2)).astype(np.float32)
    b = np.concatenate([b, b + rng.uniform(0, 300, (30, 2)).astype(np.float32)], 1)
    b[:5] = a[:5]                                             # identical boxes: IoU exactly 1
    got = oracle.box_iou(a, b)
    want = np.array([[_np_iou(x, y) for y in b] for x in a], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(got[np.arange(5), np.arange(5)] == 1.0)
    # FPN level = clamp(floor(4 + log2(sqrt(w h) / 224)), 2, 5) with the +1 box convention, thresholds on the area
    rois = np.concatenate([np.zeros((40, 1), np.float32), a], 1)
    area = (a[:, 2] - a[:, 0] + _f(1)) * (a[:, 3] - a[:, 1] + _f(1))
    lv = np.clip(np.floor(4 + np.log2(np.sqrt(area.astype(np.float64)) / 224.0)), 2, 5).astype(np.int32)
    assert np.array_equal(oracle.fpn_level(rois), lv)
    # smooth-L1 with sigma (mx-rcnn smooth_l1(scalar=sigma)): 0.5 (sigma x)^2 if |x| < 1/sigma^2 else |x| - 0.5/sigma^2
    for sigma in (1.0, 3.0):
        p = (rng.standard_normal(500) * 1.5).astype(np.float32)
        t = (rng.standard_normal(500) * 1.5).astype(np.float32)
        s2 = _f(sigma * sigma)
        x = (p - t).astype(np.float32)
        inv = _f(_f(1) / s2)
        lo = (_f(_f(0.5) * s2) * x).astype(np.float32) * x
        hi = (np.abs(x) - _f(_f(0.5) * inv)).astype(np.float32)
        want_l = np.where(np.abs(x) < inv, lo, hi).astype(np.float32)
        want_g = np.where(np.abs(x) < inv, (s2 * x).astype(np.float32), np.sign(x)).astype(np.float32)
        l, g = oracle.smooth_l1(p, t, None, sigma)
        assert np.array_equal(l.view(np.uint32), want_l.view(np.uint32))
        assert np.array_equal(g.view(np.uint32), want_g.view(np.uint32))


def test_decode_encode_independent_restatement(oracle):
    rng = np.random.default_rng(321)
    CLIP = _f(4.135166556742356)                              # log(1000 / 16)
    for _ in range(200):
        x1, y1 = rng.uniform(0, 500, 2).astype(np.float32)
        w0, h0 = rng.uniform(1, 300, 2).astype(np.float32)
        box = np.array([x1, y1, _f(x1 + w0), _f(y1 + h0)], np.float32)
        d = (rng.standard_normal(4) * np.array([0.3, 0.3, 1.5, 1.5])).astype(np.float32)
        d[2:] = np.where(rng.uniform(size=2) < 0.1, 5.0, d[2:]).astype(np.float32)      # some widths beyond the clip
        im_h, im_w = _f(600), _f(800)
        w = _f(_f(box[2] - box[0]) + _f(1)); h = _f(_f(box[3] - box[1]) + _f(1))
        cx = _f(box[0] + _f(_f(0.5) * _f(w - _f(1)))); cy = _f(box[1] + _f(_f(0.5) * _f(h - _f(1))))
        dw, dh = min(d[2], CLIP), min(d[3], CLIP)
        pcx = _f(_f(d[0] * w) + cx); pcy = _f(_f(d[1] * h) + cy)
        pw = _f(oracle.expf(np.float32(dw)) * w); ph = _f(oracle.expf(np.float32(dh)) * h)
        hw = _f(_f(0.5) * _f(pw - _f(1))); hh = _f(_f(0.5) * _f(ph - _f(1)))
        o = np.array([pcx - hw, pcy - hh, pcx + hw, pcy + hh], np.float32)
        o[0::2] = np.clip(o[0::2], 0, _f(im_w - _f(1))); o[1::2] = np.clip(o[1::2], 0, _f(im_h - _f(1)))
        got = oracle.decode_clip(box, d, im_h, im_w)
        assert np.array_equal(got.view(np.uint32), o.view(np.uint32)), (box, d, got, o)
        # encode: targets of a second box relative to the first
        g1 = rng.uniform(0, 500, 2).astype(np.float32)
        gt = np.array([g1[0], g1[1], _f(g1[0] + rng.uniform(1, 300)), _f(g1[1] + rng.uniform(1, 300))], np.float32)
        gw = _f(_f(gt[2] - gt[0]) + _f(1)); gh = _f(_f(gt[3] - gt[1]) + _f(1))
        gcx = _f(gt[0] + _f(_f(0.5) * _f(gw - _f(1)))); gcy = _f(gt[1] + _f(_f(0.5) * _f(gh - _f(1))))
        want = np.array([_f(_f(gcx - cx) / w), _f(_f(gcy - cy) / h), oracle.logf(np.float32(_f(gw / w))),
                         oracle.logf(np.float32(_f(gh / h)))], np.float32)
        got = oracle.encode(box, gt)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (box, gt, got, want)
