"""GPU parity tests for the Mask R-CNN specific pieces and a Mask R-CNN step."""
import numpy as np
import pytest
import torch

from conftest import synth_boxes, synth_gt

pytestmark = pytest.mark.gpu


def _t(a, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dt is None else t.to(dt)


def ellipse_masks(gt, H, W):
    """[N,G,H,W] u8: a filled axis-aligned ellipse inside every valid GT box."""
    N, G = gt.shape[:2]
    m = np.zeros((N, G, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for n in range(N):
        for g in range(G):
            if gt[n, g, 4] < 0:
                continue
            x1, y1, x2, y2 = gt[n, g, :4]
            cx, cy, rx, ry = 0.5 * (x1 + x2), 0.5 * (y1 + y2), 0.5 * (x2 - x1) + 0.5, 0.5 * (y2 - y1) + 0.5
            m[n, g] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).astype(np.uint8)
    return m


def test_mask_target_bit_exact(hip, oracle):
    from mxdetection_amd.core.mask import mask_target
    rng = np.random.default_rng(21)
    N, G, H, W, R = 2, 6, 96, 128, 60
    gt = synth_gt(rng, N, G, H, W, 2, 6)
    masks = ellipse_masks(gt, H, W)
    rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), synth_boxes(rng, R, H, W)], 1)
    matched = rng.integers(-1, G, R).astype(np.int32)
    labels = rng.integers(-1, 5, R).astype(np.int32)
    for i in range(0, R, 3):     # some rois hugging a GT box
        n, g = int(rois[i, 0]), int(rng.integers(0, G))
        if gt[n, g, 4] >= 0:
            rois[i, 1:] = gt[n, g, :4] + rng.uniform(-3, 3, 4)
            matched[i], labels[i] = g, int(gt[n, g, 4])
    for S in (28, 14):
        tg, cls = mask_target(_t(rois), _t(matched), _t(labels), _t(masks), S)
        w_tg, w_cls = oracle.mask_target(rois, matched, labels, masks, S)
        assert np.array_equal(cls.cpu().numpy(), w_cls)
        assert np.array_equal(tg.cpu().numpy(), w_tg)
        assert 0 < w_tg[w_cls > 0].mean() < 1      # the targets are non-trivial


def test_pixel_shuffle_and_deconv_equivalence(hip, oracle):
    import torch
    from mxdetection_amd.ops import dense
    rng = np.random.default_rng(22)
    R, H, W, Cin, C = 3, 7, 5, 64, 32
    x = _t(oracle.round_bf16(rng.standard_normal((R, H, W, Cin)).astype(np.float32)), torch.bfloat16)
    # deconv weight [Cin, Cout, 2, 2] (torch layout) -> 1x1 conv filter [(dy,dx,co), 1, 1, ci]
    wd = oracle.round_bf16((rng.standard_normal((Cin, C, 2, 2)) * 0.1).astype(np.float32))
    w1 = np.ascontiguousarray(wd.transpose(2, 3, 1, 0).reshape(4 * C, 1, 1, Cin))
    y4 = dense.conv2d_forward(x, _t(w1, torch.bfloat16))
    up = dense.pixel_shuffle2(y4)
    ref = torch.nn.functional.conv_transpose2d(x.float().cpu().permute(0, 3, 1, 2), torch.from_numpy(wd), stride=2)
    ref = ref.permute(0, 2, 3, 1).numpy()
    assert tuple(up.shape) == ref.shape
    # bf16 output of a K=64 contraction: 2^-7 relative + 2^-7 of the rms
    got = up.float().cpu().numpy()
    assert np.all(np.abs(got - ref) <= 2.0 ** -7 * np.abs(ref) + 2.0 ** -7 * np.sqrt((ref ** 2).mean()))
    back = dense.pixel_shuffle2(up, inverse=True)
    assert torch.equal(back, y4)


def test_mask_loss(hip, oracle):
    import torch
    from mxdetection_amd.core import mask as M_
    rng = np.random.default_rng(23)
    R, S, Cp = 40, 28, 128
    logits = oracle.round_bf16((rng.standard_normal((R, S, S, Cp)) * 2).astype(np.float32))
    cls = rng.integers(-1, 81, R).astype(np.int32)
    cls[cls == 0] = -1
    tg = (rng.uniform(size=(R, S, S)) < 0.4).astype(np.uint8)
    loss = torch.zeros(1, device="cuda")
    grad = torch.empty((R, S, S, Cp), dtype=torch.bfloat16, device="cuda")
    ws = M_.mask_loss_workspace(R, S, "cuda")
    M_.mask_loss(_t(logits, torch.bfloat16), _t(cls), _t(tg), loss, grad, ws)
    w_loss, w_grad = oracle.mask_loss(logits, cls, tg)
    assert np.allclose(loss.cpu().numpy(), w_loss, rtol=2e-5)        # fp32 fixed-order sum vs float64
    assert np.array_equal(grad.view(torch.int16).cpu().numpy().view(np.uint16), oracle.f32_to_bf16_bits(w_grad))


def test_mask_rcnn_step(hip):
    import torch
    from mxdetection_amd.models import FasterRCNN
    rng = np.random.default_rng(24)
    N, H, W = 2, 192, 256
    gt = synth_gt(rng, N, 8, H, W - 4, 2, 5)
    masks = ellipse_masks(gt, H, W)
    img = torch.randn((N, 3, H, W), generator=torch.Generator().manual_seed(5)).cuda()
    info = torch.tensor([[H, W - 4, 1.0]] * N).cuda()
    m = FasterRCNN("cuda", seed=7, pre_nms_top_n=600, post_nms_top_n=300, rois_per_image=128, with_mask=True)
    hist = []
    for it in range(8):
        rpn, rcnn, mk = m.train_step(img, _t(gt), info, step=0, lr=0.002, gt_masks=_t(masks))
        hist.append([float(v) for v in torch.cat([rpn, rcnn, mk]).cpu()])
    hist = np.array(hist)
    assert np.all(np.isfinite(hist))
    assert 0.55 < hist[0, 4] < 0.85          # untrained sigmoid BCE ~ ln 2
    assert hist[-1, 4] < hist[0, 4]          # the mask loss goes down on a fixed batch
    assert hist[-1, 0] < hist[0, 0]          # ... and so does the RPN objectness loss
    g = m.arena.view(m.mask_head.convs[0].wi, "g")
    assert torch.isfinite(g).all() and g.abs().sum() > 0


def test_mask_paste_bit_exact(hip, oracle):
    """Inference paste-back vs the C oracle: boxes inside, clipped by the frame, one pixel wide, empty, class -1."""
    from mxdetection_amd.core import mask as M_
    rng = np.random.default_rng(11)
    R, S, Cpad, H, W = 9, 28, 88, 96, 132
    logits = oracle.round_bf16((rng.standard_normal((R, S, S, Cpad)) * 3.0).astype(np.float32))
    dets = np.zeros((R, 6), np.float32)
    dets[0] = (10.2, 5.7, 80.4, 60.1, 0.9, 3)
    dets[1] = (-20.0, -8.0, 40.0, 30.0, 0.8, 81)           # clipped by the frame top-left
    dets[2] = (100.0, 70.0, 180.0, 140.0, 0.7, 1)          # clipped bottom-right
    dets[3] = (50.0, 50.0, 50.0, 90.0, 0.6, 7)             # one pixel wide
    dets[4] = (30.5, 40.5, 31.5, 41.5, 0.5, 12)            # half-way corners: round half to even
    dets[5] = (60.0, 20.0, 40.0, 10.0, 0.4, 5)             # inverted box: nothing
    dets[6] = (0.0, 0.0, 0.0, 0.0, 0.0, -1)                # padding row
    dets[7] = (0.0, 0.0, 131.0, 95.0, 0.3, 80)             # whole frame
    dets[8] = (5.0, 5.0, 8.0, 7.0, 0.2, 2)                 # smaller than the 28x28 map (down-scaling)
    got = M_.mask_paste(torch.from_numpy(logits).cuda().to(torch.bfloat16), torch.from_numpy(dets).cuda(), H, W).cpu().numpy()
    want = oracle.mask_paste(logits, dets, H, W)
    assert np.array_equal(got, want)
    assert not got[5].any() and not got[6].any() and got[0].any() and got[7].any()
    # known answer: logits +8 on the left half of the map and -8 on the right -> the left half of the box is set
    lg = np.full((1, S, S, 8), -8.0, np.float32)
    lg[:, :, :S // 2, 0] = 8.0
    d = np.array([[20, 10, 59, 49, 1.0, 1]], np.float32)
    m = M_.mask_paste(torch.from_numpy(lg).cuda().to(torch.bfloat16), torch.from_numpy(d).cuda(), 64, 64).cpu().numpy()[0]
    want = np.zeros((64, 64), np.uint8)
    want[10:50, 20:40] = 1
    assert np.array_equal(m, want)


def test_mask_rcnn_predict_with_masks(hip):
    from mxdetection_amd.models import FasterRCNN
    model = FasterRCNN("cuda", depth=50, seed=7, with_mask=True, num_classes=81)
    torch.manual_seed(0)
    img = torch.randn((1, 3, 256, 320), device="cuda")
    info = torch.tensor([[256.0, 320.0, 1.0]], device="cuda")
    dets, num, masks = model.predict(img, info, score_thresh=0.0, max_per_image=20, with_masks=True)
    torch.cuda.synchronize()
    n = int(num[0])
    assert masks.shape == (1, 20, 256, 320) and masks.dtype == torch.uint8 and n > 0
    d = dets[0].cpu().numpy()
    m = masks[0].cpu().numpy()
    assert not m[n:].any()                                 # padding rows paste nothing
    for k in range(n):                                      # every mask lives inside its (rounded) box
        ys, xs = np.nonzero(m[k])
        if ys.size:
            assert xs.min() >= np.rint(d[k, 0]) and xs.max() <= np.rint(d[k, 2]) and ys.min() >= np.rint(d[k, 1]) and ys.max() <= np.rint(d[k, 3])


def test_pixel_shuffle_inverse_with_relu_mask(hip):
    """The fused backward pass equals inverse shuffle followed by the ReLU mask, bit for bit."""
    from mxdetection_amd.ops import dense
    g = torch.Generator().manual_seed(2)
    R, H, W, C = 5, 7, 6, 24
    dy_up = torch.randn((R, 2 * H, 2 * W, C), generator=g).to(torch.bfloat16).cuda()
    act = torch.relu(torch.randn((R, H, W, 4 * C), generator=g)).to(torch.bfloat16).cuda()
    act[0, 0, 0, :8] = 0.0
    act[0, 0, 0, 8] = -0.0
    want = dense.pixel_shuffle2(dy_up, inverse=True)
    want = torch.where(act > 0, want, torch.zeros_like(want))
    got = dense.pixel_shuffle2_inv_relu(dy_up, act)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
