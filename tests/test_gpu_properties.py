"""Size-independent properties at the BASELINE.json shapes (no oracle involved: the kernels are checked against each
other and against algebra):

  * adjointness of the three convolution kernels: <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)>
    on the benchmark's own layer shapes (P2-level 3x3 with 256x256 tiles + tail, C4 3x3 / 1x1 static-tap tiles,
    a stride-2 3x3 with the parity-grouped data gradient);
  * adjointness of RoIAlign forward and the segment-form backward on the 200x336 pyramid with 1,024 rois;
  * NMS idempotence and order: NMS of the kept boxes keeps all of them; kept indices ascend (= scores descend);
  * linearity of the weight gradient in dy (split-K slabs + fold are plain sums).

The operands are bf16 and every kernel rounds its bf16 output once, so the identities hold to about 2^-8 relative to the
norms involved; the bounds below are 1e-2 of |a||b| (a wrong tap, a dropped row or a mis-masked border moves them by
percent, a transposed filter by order 1)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dot(a, b):
    return float((a.double().flatten() * b.double().flatten()).sum())


@pytest.mark.parametrize("N,H,W,Cin,Cout,K,stride", [
    (2, 200, 336, 256, 256, 3, 1),     # RPN / FPN P2 layer: 256x256 tiles + 64x64 tail, wgrad split 64 ways
    (2, 50, 84, 256, 256, 3, 1),       # C4 3x3: static-tap 64x64 tiles
    (2, 50, 84, 1024, 256, 1, 1),      # C4 1x1 reduce
    (2, 100, 168, 256, 256, 3, 2),     # stride 2: strided static forward, parity-grouped data gradient
    (1024, 1, 1, 12544, 1024, 1, 1),   # fc1
])
def test_conv_adjoint_identities(hip, N, H, W, Cin, Cout, K, stride):
    import torch
    from mxdetection_amd.ops import dense
    g = torch.Generator(device="cuda").manual_seed(N + H + Cin + K + stride)
    pad = K // 2
    x = torch.randn((N, H, W, Cin), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((Cout, K, K, Cin), device="cuda", generator=g) * (1.0 / (K * K * Cin) ** 0.5)).to(torch.bfloat16)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    dy = torch.randn((N, Ho, Wo, Cout), device="cuda", generator=g).to(torch.bfloat16)
    y = dense.conv2d_forward(x, w, None, None, stride, pad, False)
    dx = dense.conv2d_dgrad(dy, dense.filter_transpose(w), tuple(x.shape), K, K, stride, pad)
    dw = dense.conv2d_wgrad(x, dy, K, K, stride, pad)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = float(y.float().norm()) * float(dy.float().norm())
    assert abs(a - b) <= 1e-2 * scale and abs(a - c) <= 1e-2 * scale, (a, b, c, scale)
    # (<w, wgrad> accumulates in fp32 and is not rounded to bf16: it must sit much closer to the forward value
    # than the bound above whenever the forward value itself is not tiny)
    assert abs(a - c) <= 3e-3 * scale


def test_wgrad_is_linear_in_dy(hip):
    import torch
    from mxdetection_amd.ops import dense
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((2, 100, 168, 128), device="cuda", generator=g).to(torch.bfloat16)
    d1 = torch.randn((2, 100, 168, 128), device="cuda", generator=g).to(torch.bfloat16)
    d2 = (d1.float() * 2.0).to(torch.bfloat16)                 # exact in bf16
    w1 = dense.conv2d_wgrad(x, d1, 3, 3, 1, 1)
    w2 = dense.conv2d_wgrad(x, d2, 3, 3, 1, 1)
    assert torch.equal(w2, w1 * 2.0)                           # a power-of-two scale commutes with every rounding


def test_roi_align_adjoint_at_full_size(hip):
    """<RoIAlign(F), G> = <F, RoIAlign^T(G)> on the benchmark pyramid with the rois of a real step."""
    import torch
    from mxdetection_amd.ops import fpn_level_map, roi_align_backward_gather, roi_align_forward
    d = np.load(os.path.join(ROOT, "tools", "data", "bench_rois.npz"))
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    for key in ("3", "4"):                                     # the two steps with rois on every level
        rois = torch.from_numpy(d["rois" + key]).cuda()
        levels = fpn_level_map(rois)
        assert torch.equal(levels.cpu(), torch.from_numpy(d["levels" + key]))
        feats = [torch.randn((2, H, W, 256), device="cuda", generator=g).to(torch.bfloat16) for (H, W) in shapes]
        G = torch.randn((rois.shape[0], 7, 7, 256), device="cuda", generator=g).to(torch.bfloat16)
        out = roi_align_forward(feats, scales, rois, levels, (7, 7), 2)
        grads = [torch.zeros_like(f) for f in feats]
        roi_align_backward_gather(grads, scales, rois, levels, G, 2, 2, accumulate=False)
        lhs = _dot(out, G)
        rhs = sum(_dot(f, gr) for f, gr in zip(feats, grads))
        scale = float(out.float().norm()) * float(G.float().norm())
        assert abs(lhs - rhs) <= 1e-2 * scale, (lhs, rhs, scale)


def test_nms_idempotent_and_ordered(hip):
    import torch
    from mxdetection_amd.ops import nms_batched
    rng = np.random.default_rng(5)
    B, n = 10, 2000                                            # 2 images x 5 levels, pre-NMS 2000
    c = rng.uniform(0, 1300, (B, n, 2)).astype(np.float32)
    wh = rng.uniform(8, 300, (B, n, 2)).astype(np.float32)
    boxes = torch.from_numpy(np.concatenate([c, c + wh], 2)).cuda()
    counts = torch.full((B,), n, dtype=torch.int32).cuda()
    keep, num = nms_batched(boxes, counts, 0.7)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    kept = torch.zeros_like(boxes)
    for b in range(B):
        k = keep[b, :num[b]]
        assert num[b] > 50 and np.all(np.diff(k) > 0)          # ascending input index = descending score
        kept[b, :num[b]] = boxes[b, torch.from_numpy(k).long().cuda()]
    keep2, num2 = nms_batched(kept, torch.from_numpy(num).cuda(), 0.7)
    assert np.array_equal(num2.cpu().numpy(), num)             # nothing kept suppresses anything kept
    for b in range(B):
        assert np.array_equal(keep2.cpu().numpy()[b, :num[b]], np.arange(num[b]))
