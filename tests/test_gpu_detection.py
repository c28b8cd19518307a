"""GPU parity tests: HIP detection ops (through the C-ABI) vs the C oracle on the same seeded inputs.

Bar: integer/index outputs and box coordinates bit-exact; scalar losses within the tolerance written
in each test. The oracle is "parity unpinned" by the reference (no reference code exists).
"""
import numpy as np
import pytest

from conftest import synth_boxes, synth_gt

pytestmark = pytest.mark.gpu


def _t(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _bf16_t(bits):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)


def _bits(t):
    import torch
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def test_box_iou_bit_exact(hip, oracle):
    from mxdetection_amd.core.bbox import bbox_overlaps
    rng = np.random.default_rng(0)
    a, b = synth_boxes(rng, 3000), synth_boxes(rng, 37)
    b[0] = a[0]
    got = bbox_overlaps(_t(a), _t(b)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oracle.box_iou(a, b).view(np.uint32))
    # empty input is legal
    import torch
    assert bbox_overlaps(_t(a), torch.empty((0, 4), device="cuda")).shape == (3000, 0)


def test_generate_anchors_bit_exact(hip, oracle):
    from mxdetection_amd.core.anchor import generate_anchors, generate_base_anchors
    for stride, (H, W) in zip((4, 8, 16, 32, 64), ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21))):
        base = generate_base_anchors(stride)
        got = generate_anchors(_t(base), H, W, stride).cpu().numpy()
        assert np.array_equal(got, oracle.grid_anchors(oracle.base_anchors(stride), H, W, stride))


def test_fpn_level_map(hip, oracle):
    from mxdetection_amd.ops import fpn_level_map
    rng = np.random.default_rng(3)
    b = synth_boxes(rng, 5000)
    edge = np.array([[0, 0, 110, 110], [0, 0, 111, 111], [0, 0, 223, 223], [0, 0, 447, 447], [0, 0, 0, 0]], np.float32)
    rois = np.concatenate([np.zeros((len(b) + len(edge), 1), np.float32), np.concatenate([b, edge])], 1)
    got = fpn_level_map(_t(rois)).cpu().numpy()
    assert np.array_equal(got, oracle.fpn_level(rois))


@pytest.mark.parametrize("n,thresh", [(2000, 0.7), (1000, 0.5), (65, 0.3), (64, 0.7), (1, 0.7), (4096, 0.7)])
def test_nms_batched_bit_exact(hip, oracle, n, thresh):
    from mxdetection_amd.ops import nms_batched
    rng = np.random.default_rng(n)
    B = 6
    boxes = np.stack([synth_boxes(rng, n) for _ in range(B)])
    # cluster the boxes so suppression actually happens
    boxes[:, n // 2:] = boxes[:, : n - n // 2] + rng.uniform(-6, 6, (B, n - n // 2, 4)).astype(np.float32)
    counts = np.array([n, n, max(n - 7, 0), n // 2, 0, n], np.int32)
    invalid = (rng.uniform(size=(B, n)) < 0.05).astype(np.uint8)
    keep, num = nms_batched(_t(boxes), _t(counts), thresh, invalid=_t(invalid))
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for b in range(B):
        want = oracle.nms(boxes[b, : counts[b]], thresh, invalid=invalid[b, : counts[b]])
        assert num[b] == len(want)
        assert np.array_equal(keep[b, : num[b]], want)
    # max_keep truncates but does not change which boxes come first
    keep2, num2 = nms_batched(_t(boxes), _t(counts), thresh, max_keep=5, invalid=_t(invalid))
    assert np.array_equal(keep2.cpu().numpy()[0, :5], keep[0, :5]) and int(num2[0]) == min(5, num[0])


def _pyramid_inputs(rng, N, shapes, A, dtype_bf16, oracle):
    scores, deltas, fused = [], [], []
    for (H, W) in shapes:
        s = rng.standard_normal((N, H, W, A)).astype(np.float32) * 2.0
        d = (rng.standard_normal((N, H, W, A, 4)) * 0.3).astype(np.float32)
        if dtype_bf16:
            s, d = oracle.round_bf16(s), oracle.round_bf16(d)   # forces score ties
        scores.append(s.reshape(N, -1))
        deltas.append(d.reshape(N, -1, 4))
        f = np.zeros((N, H, W, 16), np.float32)
        f[..., :A] = s
        f[..., A:5 * A] = d.reshape(N, H, W, 4 * A)
        fused.append(f)
    return scores, deltas, fused


@pytest.mark.parametrize("bf16", [True, False])
def test_proposal_bit_exact(hip, oracle, bf16):
    import torch
    from mxdetection_amd.core.anchor import generate_base_anchors
    from mxdetection_amd.ops import PyramidProposal
    rng = np.random.default_rng(11)
    N, A = 2, 3
    shapes = [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)]
    strides = [4, 8, 16, 32, 64]
    scores, deltas, fused = _pyramid_inputs(rng, N, shapes, A, bf16, oracle)
    im_info = np.array([[192, 320, 1.0], [180, 300, 1.0]], np.float32)
    base = [generate_base_anchors(s) for s in strides]
    pre, post = 600, 500
    op = PyramidProposal([_t(b) for b in base], strides, pre, post, 0.7, 4.0)
    dt = torch.bfloat16 if bf16 else torch.float32
    cls = [_t(f, dt) for f in fused]
    rois, sc, anc, num = op(cls, cls, _t(im_info))
    w_rois, w_sc, w_anc, w_num = oracle.proposal(scores, deltas, base, [s[0] for s in shapes], [s[1] for s in shapes],
                                                 strides, im_info, pre, post, 0.7, 4.0)
    assert np.array_equal(num.cpu().numpy(), w_num)
    assert np.array_equal(anc.cpu().numpy(), w_anc)
    assert np.array_equal(rois.cpu().numpy().view(np.uint32), w_rois.view(np.uint32))
    assert np.array_equal(sc.cpu().numpy().view(np.uint32), w_sc.view(np.uint32))
    # NCHW-strided producers go through the same entry point
    if not bf16:
        cls_nchw = [_t(np.ascontiguousarray(s.reshape(N, H, W, A).transpose(0, 3, 1, 2))) for s, (H, W) in zip(scores, shapes)]
        reg_nchw = [_t(np.ascontiguousarray(d.reshape(N, H, W, 4 * A).transpose(0, 3, 1, 2))) for d, (H, W) in zip(deltas, shapes)]
        r2, _, a2, n2 = op(cls_nchw, reg_nchw, _t(im_info), layout="nchw")
        assert np.array_equal(a2.cpu().numpy(), w_anc) and np.array_equal(r2.cpu().numpy(), w_rois)


def test_anchor_target_bit_exact(hip, oracle):
    from mxdetection_amd.core.anchor import assign_anchor
    rng = np.random.default_rng(5)
    strides, shapes = [4, 8, 16, 32, 64], [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    anchors = np.concatenate([oracle.grid_anchors(oracle.base_anchors(s), H, W, s) for s, (H, W) in zip(strides, shapes)])
    N = 3
    gt = synth_gt(rng, N, 16, 400, 666)
    gt[2, :, 4] = -1           # an image without any GT: everything inside becomes background
    im_info = np.array([[400, 666, 1.0]] * N, np.float32)
    for batch in (256, 0):
        lab, mg, tg, mi = assign_anchor(_t(anchors), _t(gt), _t(im_info), 0.7, 0.3, 0.0, batch, 0.5, 99, 7, 4)
        w_lab, w_mg, w_tg, w_mi = oracle.anchor_target(anchors, gt, im_info, 0.7, 0.3, 0.0, batch, 0.5, 99, 7, 4)
        assert np.array_equal(lab.cpu().numpy(), w_lab)
        assert np.array_equal(mi.cpu().numpy().view(np.uint32), w_mi.view(np.uint32))
        fg = w_lab == 1
        assert np.array_equal(mg.cpu().numpy()[fg], w_mg[fg])
        assert np.array_equal(tg.cpu().numpy().view(np.uint32), w_tg.view(np.uint32))
        if batch:
            assert all((w_lab[n] >= 0).sum() <= batch and (w_lab[n] == 1).sum() <= batch // 2 for n in range(N))


def test_proposal_target_bit_exact(hip, oracle):
    from mxdetection_amd.core.bbox import sample_rois
    rng = np.random.default_rng(6)
    N, S, G, R = 3, 700, 12, 128
    gt = synth_gt(rng, N, G, 400, 666, 2, 10)
    rois = np.zeros((N, S, 5), np.float32)
    for n in range(N):
        b = synth_boxes(rng, S, 400, 666)
        k = 0
        for g in gt[n]:
            if g[4] >= 0:
                for _ in range(25):   # jittered copies of GT so there are foreground candidates
                    b[k] = g[:4] + rng.uniform(-8, 8, 4)
                    k += 1
        rois[n, :, 0] = n
        rois[n, :, 1:] = b
    num_rois = np.array([S, S - 100, 10], np.int32)
    gt[2, :, 4] = -1
    got = sample_rois(_t(rois), _t(num_rois), _t(gt), R, 0.25, 0.5, 0.5, 0.0, 81, False, (0, 0, 0, 0),
                      (0.1, 0.1, 0.2, 0.2), 99, 3, 8)
    want = oracle.proposal_target(rois, num_rois, gt, R, 0.25, 0.5, 0.5, 0.0, 81, False, (0, 0, 0, 0),
                                  (0.1, 0.1, 0.2, 0.2), 99, 3, 8)
    names = ["rois", "labels", "targets", "weights", "matched", "num_fg"]
    for g_, w_, nm in zip(got, want, names):
        g_ = g_.cpu().numpy()
        assert np.array_equal(g_.view(np.uint32) if g_.dtype == np.float32 else g_,
                              w_.view(np.uint32) if w_.dtype == np.float32 else w_), nm


def _feat_pyramid(rng, N, C, shapes, oracle):
    return [oracle.f32_to_bf16_bits(rng.standard_normal((N, H, W, C)).astype(np.float32)) for (H, W) in shapes]


def test_roi_align_forward_bit_exact(hip, oracle):
    from mxdetection_amd.ops import fpn_level_map, roi_align_forward
    rng = np.random.default_rng(8)
    N, C = 2, 64
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    feats = _feat_pyramid(rng, N, C, shapes, oracle)
    R = 300
    b = synth_boxes(rng, R, 400, 666)
    b[:5] = [[0, 0, 0, 0], [660, 395, 665, 399], [-20, -20, 5, 5], [600, 300, 900, 700], [10, 10, 10.5, 10.5]]
    rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
    levels = fpn_level_map(_t(rois))
    assert np.array_equal(levels.cpu().numpy(), oracle.fpn_level(rois))
    out = roi_align_forward([_bf16_t(f) for f in feats], scales, _t(rois), levels, (7, 7), 2)
    want = oracle.roi_align(feats, scales, rois, levels.cpu().numpy(), 7, 7, 2)
    assert np.array_equal(_bits(out), want)
    # adaptive sampling grid (sampling_ratio = 0) and 14x14 pooling (mask branch geometry)
    out = roi_align_forward([_bf16_t(f) for f in feats], scales, _t(rois[:64]), levels[:64], (14, 14), 0)
    want = oracle.roi_align(feats, scales, rois[:64], levels.cpu().numpy()[:64], 14, 14, 0)
    assert np.array_equal(_bits(out), want)


def test_roi_align_backward_tolerance(hip, oracle):
    import torch
    from mxdetection_amd.ops import roi_align_backward
    rng = np.random.default_rng(9)
    N, C = 2, 32
    shapes = [(50, 84), (25, 42)]
    scales = [0.125, 0.0625]
    feats = _feat_pyramid(rng, N, C, shapes, oracle)
    R = 100
    rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), synth_boxes(rng, R, 400, 666)], 1)
    levels = rng.integers(3, 5, R).astype(np.int32)
    go = oracle.f32_to_bf16_bits(rng.standard_normal((R, 7, 7, C)).astype(np.float32))
    dfe = [torch.zeros((N, H, W, C), dtype=torch.float32, device="cuda") for (H, W) in shapes]
    roi_align_backward(dfe, scales, _t(rois), _t(levels), _bf16_t(go), 2, 3)
    want = oracle.roi_align(feats, scales, rois, levels, 7, 7, 2, 3, grad_out_bits=go)
    for g, w in zip(dfe, want):
        # fp32 atomics: summation order differs run to run -> tolerance 1e-5 relative to the map's scale
        assert np.allclose(g.cpu().numpy(), w, rtol=1e-5, atol=1e-5 * np.abs(w).max())


def test_roi_align_backward_gather_form(hip, oracle):
    """The deterministic gather form against the oracle (tolerance: its fixed summation order is not the oracle's) and
    against itself (two runs bit-identical); written straight into bf16 maps, with and without accumulation."""
    import torch
    from mxdetection_amd.ops import roi_align_backward_gather
    rng = np.random.default_rng(19)
    N, C = 2, 40                       # C not a multiple of 64: lanes past C idle
    shapes = [(50, 84), (25, 42), (13, 21)]
    scales = [0.125, 0.0625, 0.03125]
    feats = _feat_pyramid(rng, N, C, shapes, oracle)
    R = 160
    b = synth_boxes(rng, R, 400, 666)
    b[:5] = [[0, 0, 0, 0], [660, 395, 665, 399], [-20, -20, 5, 5], [600, 300, 900, 700], [10, 10, 10.5, 10.5]]
    rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
    levels = rng.integers(3, 6, R).astype(np.int32)
    go = oracle.f32_to_bf16_bits(rng.standard_normal((R, 7, 7, C)).astype(np.float32))
    want = oracle.roi_align(feats, scales, rois, levels, 7, 7, 2, 3, grad_out_bits=go)
    maps = [torch.full((N, H, W, C), 7.0, dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]   # must be overwritten
    roi_align_backward_gather(maps, scales, _t(rois), _t(levels), _bf16_t(go), 2, 3, accumulate=False)
    first = [m.clone() for m in maps]
    for g, w in zip(maps, want):
        w16 = oracle.round_bf16(w)
        # fp32 sums in a different fixed order (per-bin collapsed weights), then one bf16 rounding: 1e-5 of the map's
        # scale + one bf16 ulp (a last-bit fp32 difference can flip the rounding of a value that sits on a tie)
        assert np.allclose(g.float().cpu().numpy(), w16, rtol=2.0 ** -7, atol=1e-5 * np.abs(w).max())
    roi_align_backward_gather(maps, scales, _t(rois), _t(levels), _bf16_t(go), 2, 3, accumulate=False)
    for a, b2 in zip(first, maps):
        assert torch.equal(a, b2)                                   # deterministic
    base = [torch.from_numpy(oracle.round_bf16(rng.standard_normal((N, H, W, C)).astype(np.float32))).cuda().to(torch.bfloat16)
            for (H, W) in shapes]
    acc = [t.clone() for t in base]
    roi_align_backward_gather(acc, scales, _t(rois), _t(levels), _bf16_t(go), 2, 3, accumulate=True)
    for a, b0, w in zip(acc, base, want):
        ref = b0.float().cpu().numpy() + w
        assert np.allclose(a.float().cpu().numpy(), ref, rtol=2.0 ** -7, atol=2e-5 * np.abs(w).max() + 2.0 ** -8 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("PH,PW,sr,C,R", [(7, 7, 2, 256, 700), (14, 14, 2, 72, 300), (7, 7, 1, 64, 300), (7, 7, 3, 64, 200),
                                          (5, 9, 2, 320, 1500)])
def test_roi_align_backward_segment_form_shapes(hip, oracle, PH, PW, sr, C, R):
    """Segment form (one launch; per-bin collapsed weights) on the shapes the table form never saw: 256+ channels (two
    channel blocks), the mask branch's 14x14 bins, sampling ratios 1 and 3 (count 9: exact divisions), more rois than one
    list round, rois outside the map, one-pixel rois. Against the oracle with the fp32-reassociation tolerance, against
    the table form, and twice against itself (bit-identical)."""
    import torch
    from mxdetection_amd import _lib
    from mxdetection_amd.ops import roi_align_backward_gather
    rng = np.random.default_rng(100 + PH + sr + C)
    N = 2
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    feats = [np.zeros((N, H, W, C), dtype=np.uint16) for (H, W) in shapes]
    b = synth_boxes(rng, R, 400, 666)
    b[:6] = [[0, 0, 0, 0], [660, 395, 665, 399], [-20, -20, 5, 5], [600, 300, 900, 700], [10, 10, 10.5, 10.5],
             [-500, -500, -450, -440]]
    rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
    levels = rng.integers(2, 6, R).astype(np.int32)
    go = oracle.f32_to_bf16_bits(rng.standard_normal((R, PH, PW, C)).astype(np.float32))
    want = oracle.roi_align(feats, scales, rois, levels, PH, PW, sr, 2, grad_out_bits=go)
    maps = [torch.full((N, H, W, C), 7.0, dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]
    roi_align_backward_gather(maps, scales, _t(rois), _t(levels), _bf16_t(go), sr, 2, accumulate=False)
    first = [m.clone() for m in maps]
    for g, w in zip(maps, want):
        w16 = oracle.round_bf16(w)
        assert np.allclose(g.float().cpu().numpy(), w16, rtol=2.0 ** -7, atol=2e-5 * np.abs(w).max())
    roi_align_backward_gather(maps, scales, _t(rois), _t(levels), _bf16_t(go), sr, 2, accumulate=False)
    for a, b2 in zip(first, maps):
        assert torch.equal(a, b2)
    # records written ahead of the gather (as the training step does in its forward pass): same bits
    from mxdetection_amd.ops.roi_align import roi_align_backward_gather_prepare, roi_align_backward_gather_workspace
    ws = roi_align_backward_gather_workspace(maps, scales, R, 2)
    roi_align_backward_gather_prepare(maps, scales, _t(rois), _t(levels), (PH, PW), sr, 2, ws)
    pre = [torch.full((N, H, W, C), 3.0, dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]
    roi_align_backward_gather(pre, scales, _t(rois), _t(levels), _bf16_t(go), sr, 2, accumulate=False, workspace=ws, prepared=True)
    for a, b2 in zip(first, pre):
        assert torch.equal(a, b2)
    lib = _lib.load()
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["ROI_TABLE"], 1)
    try:
        tab = [torch.full((N, H, W, C), 7.0, dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]
        roi_align_backward_gather(tab, scales, _t(rois), _t(levels), _bf16_t(go), sr, 2, accumulate=False)
    finally:
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["ROI_TABLE"], -1)
    for a, t, w in zip(first, tab, want):
        assert np.allclose(a.float().cpu().numpy(), t.float().cpu().numpy(), rtol=2.0 ** -7, atol=2e-5 * np.abs(w).max())
    # accumulate: untouched pixels keep their bits
    base = [torch.from_numpy(oracle.round_bf16(rng.standard_normal((N, H, W, C)).astype(np.float32))).cuda().to(torch.bfloat16)
            for (H, W) in shapes]
    acc = [t.clone() for t in base]
    roi_align_backward_gather(acc, scales, _t(rois), _t(levels), _bf16_t(go), sr, 2, accumulate=True)
    for a, b0, w in zip(acc, base, want):
        ref = b0.float().cpu().numpy() + w
        assert np.allclose(a.float().cpu().numpy(), ref, rtol=2.0 ** -7, atol=2e-5 * np.abs(w).max() + 2.0 ** -8 * np.abs(ref).max())
        untouched = (w == 0).all(axis=-1)
        assert torch.equal(a[torch.from_numpy(untouched).cuda()], b0[torch.from_numpy(untouched).cuda()])


def test_rpn_loss_level(hip, oracle):
    import torch
    from mxdetection_amd.core import loss as L
    rng = np.random.default_rng(10)
    N, H, W, A, Cp = 2, 25, 42, 3, 16
    A_total, off = 2 * H * W * A + 100, 100
    head = oracle.round_bf16((rng.standard_normal((N, H, W, Cp)) * 1.5).astype(np.float32))
    labels = rng.choice([-1, -1, -1, 0, 1], size=(N, A_total)).astype(np.int32)
    targets = (rng.standard_normal((N, A_total, 4)) * 0.5).astype(np.float32)
    norm, scale = 1.0 / 512, 1.0
    nparts = L.rpn_loss_num_partials(N, H, W)
    partial = torch.zeros((nparts * 2,), dtype=torch.float32, device="cuda")
    grad = torch.empty((N, H, W, Cp), dtype=torch.bfloat16, device="cuda")
    L.rpn_loss_level(_t(head, torch.bfloat16), A, _t(labels), _t(targets), off, 3.0, norm, scale, grad, partial)
    out = torch.empty((2,), dtype=torch.float32, device="cuda")
    L.loss_finalize(partial, nparts, 2, out)
    w_loss, w_grad = oracle.rpn_loss_level(head, A, labels, targets, off, 3.0, norm, scale)
    # losses: fp32 fixed-order sum vs float64 sum -> 1e-5 relative
    assert np.allclose(out.cpu().numpy(), w_loss, rtol=1e-5)
    # gradients: identical fp32 formula, then bf16 rounding -> bit-exact
    assert np.array_equal(_bits(grad), oracle.f32_to_bf16_bits(w_grad))
    # determinism: a second run gives the same bits
    L.rpn_loss_level(_t(head, torch.bfloat16), A, _t(labels), _t(targets), off, 3.0, norm, scale, grad, partial)
    out2 = torch.empty_like(out)
    L.loss_finalize(partial, nparts, 2, out2)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("bf16", [False, True])
def test_rcnn_loss(hip, oracle, bf16):
    import torch
    from mxdetection_amd.core import loss as L
    rng = np.random.default_rng(12)
    R, NC = 300, 81
    ld = 416
    fused = (rng.standard_normal((R, ld)) * 2).astype(np.float32)
    if bf16:
        fused = oracle.round_bf16(fused)
    labels = rng.integers(-1, NC, R).astype(np.int32)
    tgt = np.zeros((R, 4 * NC), np.float32)
    wgt = np.zeros((R, 4 * NC), np.float32)
    for r in range(R):
        if labels[r] > 0:
            tgt[r, 4 * labels[r]: 4 * labels[r] + 4] = rng.standard_normal(4)
            wgt[r, 4 * labels[r]: 4 * labels[r] + 4] = 1.0
    dt = torch.bfloat16 if bf16 else torch.float32
    x = _t(fused, dt)
    g = torch.zeros_like(x)
    out = torch.empty((2,), dtype=torch.float32, device="cuda")
    ws = L.loss_workspace(R, "cuda")
    L.rcnn_loss(x, x[:, NC:], _t(labels), _t(tgt), _t(wgt), NC, 4 * NC, ld, ld, 1.0, 1.0 / R, 1.0, g, g[:, NC:], out, ws)
    w_loss, w_gc, w_gr = oracle.rcnn_loss(fused[:, :NC], fused[:, NC:NC + 4 * NC], labels, tgt, wgt, NC, 4 * NC, 1.0,
                                          1.0 / R, 1.0)
    assert np.allclose(out.cpu().numpy(), w_loss, rtol=2e-5)
    gg = g.float().cpu().numpy()
    tol = 1e-2 if bf16 else 1e-6   # bf16 gradient storage: 2^-8 relative
    assert np.allclose(gg[:, :NC], w_gc, rtol=tol, atol=tol * 1e-3)
    assert np.allclose(gg[:, NC:NC + 4 * NC], w_gr, rtol=tol, atol=tol * 1e-3)
    assert np.all(gg[:, NC + 4 * NC:] == 0)


def test_focal_loss(hip, oracle):
    import torch
    from mxdetection_amd.core.loss import focal_loss
    rng = np.random.default_rng(13)
    n, C = 5000, 80
    logits = (rng.standard_normal((n, C)) * 3).astype(np.float32)
    labels = rng.choice(np.arange(-1, C + 1), size=n, p=[0.1, 0.8] + [0.1 / C] * C).astype(np.int32)
    loss, grad = focal_loss(_t(logits), _t(labels), 0.25, 2.0)
    w_loss, w_grad = oracle.focal_loss(logits, labels, 0.25, 2.0)
    assert np.allclose(loss.cpu().numpy(), w_loss, rtol=2e-5)                 # fp32 vs float64 oracle
    assert np.allclose(grad.cpu().numpy(), w_grad, rtol=1e-4, atol=1e-8)
    lb, gb = focal_loss(_t(oracle.round_bf16(logits), torch.bfloat16), _t(labels), 0.25, 2.0)
    w_loss, w_grad = oracle.focal_loss(oracle.round_bf16(logits), labels, 0.25, 2.0)
    assert np.allclose(lb.cpu().numpy(), w_loss, rtol=2e-5)
    assert np.allclose(gb.float().cpu().numpy(), w_grad, rtol=1e-2, atol=1e-8)   # bf16 gradient storage


def test_smooth_l1(hip, oracle):
    from mxdetection_amd.core.loss import smooth_l1, smooth_l1_backward
    rng = np.random.default_rng(14)
    p, t, w = [rng.standard_normal(10000).astype(np.float32) for _ in range(3)]
    for sigma in (1.0, 3.0):
        out = smooth_l1(_t(p), _t(t), _t(w), sigma).cpu().numpy()
        g = smooth_l1_backward(_t(p), _t(t), _t(w), None, sigma).cpu().numpy()
        w_out, w_g = oracle.smooth_l1(p, t, w, sigma)
        assert np.array_equal(out, w_out) and np.array_equal(g, w_g)


def test_errors_are_reported_not_thrown(hip):
    import torch
    from mxdetection_amd._lib import MxdetError
    from mxdetection_amd.ops import nms_batched
    boxes = torch.zeros((1, 5000, 4), device="cuda")
    with pytest.raises(MxdetError, match="n_max"):
        nms_batched(boxes, torch.tensor([5000], dtype=torch.int32, device="cuda"), 0.5)


def test_against_committed_golden_vectors(hip):
    """HIP ops reproduce tests/golden/det_small.npz bit for bit (vectors made by tests/golden/make_golden.py)."""
    import os
    import torch
    from mxdetection_amd.core.anchor import assign_anchor, generate_base_anchors
    from mxdetection_amd.core.bbox import sample_rois
    from mxdetection_amd.ops import PyramidProposal, nms_batched, roi_align_forward
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_small.npz"))
    b = G["nms_boxes"]
    for thr, key in ((0.5, "nms_keep_0p5"), (0.7, "nms_keep_0p7")):
        keep, num = nms_batched(_t(b[None]), _t(np.array([len(b)], np.int32)), thr)
        assert np.array_equal(keep.cpu().numpy()[0, : int(num[0])], G[key])
    strides, shapes, A, N = [8, 16], [(12, 16), (6, 8)], 3, 2
    base = [generate_base_anchors(s) for s in strides]
    fused = []
    for l, (H, W) in enumerate(shapes):
        f = np.zeros((N, H, W, 16), np.float32)
        f[..., :A] = G["prop_sc%d" % l].reshape(N, H, W, A)
        f[..., A:5 * A] = G["prop_dl%d" % l].reshape(N, H, W, 4 * A)
        fused.append(_t(f, torch.bfloat16))
    op = PyramidProposal([_t(x) for x in base], strides, 100, 60, 0.7, 2.0)
    rois, sc, anc, num = op(fused, fused, _t(G["prop_info"]))
    assert np.array_equal(rois.cpu().numpy(), G["prop_rois"]) and np.array_equal(anc.cpu().numpy(), G["prop_anchor"])
    assert np.array_equal(num.cpu().numpy(), G["prop_num"]) and np.array_equal(sc.cpu().numpy(), G["prop_scores"])
    lab, _, tg, mi = assign_anchor(_t(G["at_anchors"]), _t(G["at_gt"]), _t(G["prop_info"]), 0.7, 0.3, 0.0, 64, 0.5, 99, 1, 0)
    assert np.array_equal(lab.cpu().numpy(), G["at_labels"]) and np.array_equal(tg.cpu().numpy(), G["at_targets"])
    out = sample_rois(_t(G["pt_rois_in"]), _t(np.array([60, 45], np.int32)), _t(G["at_gt"]), 32, 0.25, 0.5, 0.5, 0.0, 81,
                      False, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 99, 1, 0)
    assert np.array_equal(out[0].cpu().numpy(), G["pt_rois"]) and np.array_equal(out[1].cpu().numpy(), G["pt_labels"])
    assert np.array_equal(out[2].cpu().numpy(), G["pt_targets"])
    o = roi_align_forward([_bf16_t(G["ra_f0"]), _bf16_t(G["ra_f1"])], [1 / 8, 1 / 16], _t(G["ra_rois"]), _t(G["ra_levels"]),
                          (7, 7), 2, 3)
    assert np.array_equal(_bits(o), G["ra_out"])
