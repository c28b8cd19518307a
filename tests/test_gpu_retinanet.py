"""RetinaNet (BASELINE.json config 5) pieces: fused focal + box loss of a level vs the oracle, and a train step."""
import numpy as np
import pytest

from conftest import synth_gt

pytestmark = pytest.mark.gpu


def _t(a, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dt is None else t.to(dt)


def test_retina_loss_level_vs_oracle(hip, oracle):
    import torch
    from mxdetection_amd.core import loss as L
    rng = np.random.default_rng(31)
    N, H, W, A, C = 2, 9, 11, 9, 80
    ld_cls, ld_reg = 768, 64
    At, off = N and (H * W * A + 50), 50
    cls = oracle.round_bf16((rng.standard_normal((N, H, W, ld_cls)) * 2 - 2).astype(np.float32))
    reg = oracle.round_bf16((rng.standard_normal((N, H, W, ld_reg)) * 0.5).astype(np.float32))
    labels = rng.choice(np.arange(-1, C + 1), size=(N, At), p=[0.1, 0.8] + [0.1 / C] * C).astype(np.int32)
    targets = (rng.standard_normal((N, At, 4)) * 0.3).astype(np.float32)
    nfg = int((labels > 0).sum())          # the kernel normalises by the device word it is given
    num_fg = torch.tensor([nfg], dtype=torch.int32, device="cuda")
    nparts = L.retina_loss_num_partials(N, H, W, A)
    partial = torch.zeros(2 * nparts, device="cuda")
    gc = torch.zeros((N, H, W, ld_cls), dtype=torch.bfloat16, device="cuda")
    gr = torch.zeros((N, H, W, ld_reg), dtype=torch.bfloat16, device="cuda")
    L.retina_loss_level(_t(cls, torch.bfloat16), _t(reg, torch.bfloat16), A, C, _t(labels), _t(targets), off, 0.25, 2.0, 3.0,
                        num_fg, 1.0, gc, gr, partial)
    out = torch.zeros(2, device="cuda")
    L.loss_finalize(partial, nparts, 2, out)
    # oracle: gather this level's rows, dense focal loss + smooth-L1, renormalise to the global fg count
    lv = labels[:, off:off + H * W * A].reshape(-1)
    logits = cls[..., :A * C].reshape(-1, C)
    w_loss, w_grad = oracle.focal_loss(logits, lv, 0.25, 2.0)
    lvl_fg = max(1, int((lv > 0).sum()))
    scale = lvl_fg / max(1, nfg)
    assert np.allclose(out.cpu().numpy()[0], w_loss[0] * scale, rtol=3e-5)
    g = gc.float().cpu().numpy()[..., :A * C].reshape(-1, C)
    assert np.allclose(g, w_grad * scale, rtol=1e-2, atol=1e-7)              # bf16 gradient storage
    assert np.all(gc.float().cpu().numpy()[..., A * C:] == 0)
    d = reg[..., :4 * A].reshape(-1, 4)
    tv = targets[:, off:off + H * W * A].reshape(-1, 4)
    fg = lv > 0
    l1, g1 = oracle.smooth_l1(d[fg], tv[fg], None, 3.0)
    assert np.allclose(out.cpu().numpy()[1], l1.sum() / max(1, nfg), rtol=3e-5)
    gg = gr.float().cpu().numpy()[..., :4 * A].reshape(-1, 4)
    assert np.allclose(gg[fg], g1 / max(1, nfg), rtol=1e-2, atol=1e-7) and np.all(gg[~fg] == 0)


def test_anchor_class_labels(hip):
    import torch
    from mxdetection_amd.core import loss as L
    rng = np.random.default_rng(32)
    N, A, G = 2, 1000, 7
    labels = rng.integers(-1, 2, (N, A)).astype(np.int32)
    matched = rng.integers(0, G, (N, A)).astype(np.int32)
    gt = np.zeros((N, G, 5), np.float32)
    gt[..., 4] = rng.integers(1, 81, (N, G))
    out = torch.empty((N, A), dtype=torch.int32, device="cuda")
    nfg = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.anchor_class_labels(_t(labels), _t(matched), _t(gt), out, nfg)
    want = np.where(labels == 1, gt[np.arange(N)[:, None], matched, 4].astype(np.int32), labels)
    assert np.array_equal(out.cpu().numpy(), want) and int(nfg) == int((labels == 1).sum())


def test_retinanet_train_step(hip):
    import torch
    from mxdetection_amd.models import RetinaNet
    rng = np.random.default_rng(33)
    N, H, W = 2, 256, 320
    gt = synth_gt(rng, N, 8, H, W - 4, 2, 6)
    img = torch.randn((N, 3, H, W), generator=torch.Generator().manual_seed(9)).cuda()
    info = torch.tensor([[H, W - 4, 1.0]] * N).cuda()
    m = RetinaNet("cuda", depth=50, seed=7)
    hist = []
    for it in range(8):
        (loss,) = m.train_step(img, _t(gt), info, step=0, lr=0.002)
        hist.append(loss.cpu().numpy().copy())
    hist = np.array(hist)
    assert np.all(np.isfinite(hist)), hist
    assert hist[0, 0] > 0 and hist[0, 1] > 0
    assert hist[-1].sum() < hist[0].sum(), hist
    assert 9 * (32 * 40 + 16 * 20 + 8 * 10 + 4 * 5 + 2 * 3) == m.head.anchors.shape[0]


def test_retina_detect_bit_exact(hip, oracle):
    """mxdet_retina_detect vs the oracle restatement: 3 levels, 3 anchors x 5 classes, bf16-valued logits (heavy ties),
    padded channel dimensions as the head produces them."""
    import torch
    from mxdetection_amd.core.evaluation import RetinaDetect
    rng = np.random.default_rng(21)
    N, A, Cn = 2, 3, 5
    shapes, strides = [(16, 20), (8, 10), (4, 5)], [8, 16, 32]
    ld_cls, ld_reg = 64, 64
    base = [oracle.base_anchors(s) for s in strides]
    cls, reg, cls_o, reg_o = [], [], [], []
    for (H, W) in shapes:
        c = oracle.round_bf16((rng.standard_normal((N, H, W, ld_cls)) * 2.0 - 1.0).astype(np.float32))
        r = oracle.round_bf16((rng.standard_normal((N, H, W, ld_reg)) * 0.3).astype(np.float32))
        cls.append(torch.from_numpy(c).cuda().to(torch.bfloat16))
        reg.append(torch.from_numpy(r).cuda().to(torch.bfloat16))
        cls_o.append(c[..., :A * Cn].reshape(N, -1))
        reg_o.append(r[..., :A * 4].reshape(N, H * W * A, 4))
    info = np.array([[128, 160, 1.0], [120, 150, 1.0]], np.float32)
    det = RetinaDetect(Cn, strides, [torch.from_numpy(b).cuda() for b in base], pre_nms_top_n=60, score_thresh=0.05,
                       nms_thresh=0.5, max_per_image=30)
    dets, num = det(cls, reg, torch.from_numpy(info).cuda())
    want, wnum = oracle.retina_detect(cls_o, reg_o, base, [s[0] for s in shapes], [s[1] for s in shapes], strides, info, Cn,
                                      pre_n=60, score_thresh=0.05, nms_thresh=0.5, max_det=30)
    assert np.array_equal(num.cpu().numpy(), wnum)
    assert np.array_equal(dets.cpu().numpy().view(np.uint32), want.view(np.uint32))
    d = dets.cpu().numpy()
    for n in range(N):
        k = int(wnum[n])
        assert k == 30 and np.all(np.diff(d[n, :k, 4]) <= 0) and np.all((d[n, :k, 5] >= 1) & (d[n, :k, 5] <= Cn))
        assert np.all(d[n, :k, 2] <= info[n, 1] - 1) and np.all(d[n, :k, 0] >= 0) and np.all(d[n, k:, 5] == -1)
    # per class, no two kept boxes overlap by more than the NMS threshold
    for c in range(1, Cn + 1):
        b = d[0, :30][d[0, :30, 5] == c][:, :4]
        if len(b) > 1:
            iou = oracle.box_iou(b, b)
            assert np.all(iou[np.triu_indices(len(b), 1)] <= 0.5)


def test_retinanet_predict(hip):
    import torch
    from mxdetection_amd.models import RetinaNet
    model = RetinaNet("cuda", depth=50 if False else 101, num_classes=80, seed=3)
    torch.manual_seed(0)
    img = torch.randn((1, 3, 256, 320), device="cuda")
    info = torch.tensor([[256.0, 320.0, 1.0]], device="cuda")
    dets, num = model.predict(img, info, score_thresh=0.0, max_per_image=50)
    torch.cuda.synchronize()
    n = int(num[0])
    d = dets[0].cpu().numpy()
    assert 0 < n <= 50 and np.all(np.diff(d[:n, 4]) <= 0) and np.all(d[:n, 4] < 0.05)     # focal prior: p starts near 0.01
    assert np.all((d[:n, 5] >= 1) & (d[:n, 5] <= 80)) and np.all(d[:n, 2] <= 319) and np.all(d[:n, 3] <= 255)
