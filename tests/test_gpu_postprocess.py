"""GPU parity for test-time post-processing (core/evaluation): the C-ABI path against the C
oracle. Integer results (kept roi / class, counts) and float results (scores, boxes) are compared bit-exact: both sides
use include/mxdet_math.h and are built without FMA contraction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng, N, R, C, nvalid):
    ctr = rng.uniform(40, 600, (N * R, 2)).astype(np.float32)
    wh = np.exp(rng.uniform(np.log(16), np.log(300), (N * R, 2))).astype(np.float32)
    rois = np.zeros((N * R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(N), R)
    rois[:, 1:3] = ctr - wh / 2
    rois[:, 3:5] = ctr + wh / 2
    cls = (rng.standard_normal((N * R, C)) * 2.5).astype(np.float32)
    # a few confident foreground rois and many exact ties (bf16-like logits)
    cls[rng.integers(0, N * R, 40), rng.integers(1, C, 40)] += 8.0
    cls = np.round(cls * 8) / 8
    reg = (rng.standard_normal((N * R, 4 * C)) * 0.5).astype(np.float32)
    info = np.array([[640.0, 704.0, 1.0]] * N, np.float32)
    return cls, reg, rois, np.asarray(nvalid, np.int32), info


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_detection_postprocess_bit_exact(hip, oracle, dtype):
    import torch
    from mxdetection_amd.core.evaluation import DetectionPostprocess
    rng = np.random.default_rng(5)
    N, R, C = 2, 300, 21
    cls, reg, rois, nvalid, info = _case(rng, N, R, C, [300, 187])
    stds = (0.1, 0.1, 0.2, 0.2)
    if dtype == "bf16":
        cls, reg = oracle.round_bf16(cls), oracle.round_bf16(reg)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    # fused head layout: one [N*R, ld] tensor, cls | reg | padding
    ld = (C + 4 * C + 63) // 64 * 64
    fused = torch.zeros((N * R, ld), dtype=tdt, device="cuda")
    fused[:, :C] = torch.from_numpy(cls).to(tdt).cuda()
    fused[:, C:5 * C] = torch.from_numpy(reg).to(tdt).cuda()
    post = DetectionPostprocess(C, score_thresh=0.02, nms_thresh=0.5, max_per_image=50, stds=stds)
    dets, num = post(fused[:, :C], fused[:, C:], torch.from_numpy(rois).cuda(), torch.from_numpy(nvalid).cuda(),
                     torch.from_numpy(info).cuda())
    torch.cuda.synchronize()
    w_dets, w_num, _, _ = oracle.detection_postprocess(cls, reg, rois, nvalid, info, (0, 0, 0, 0), stds, 0.02, 0.5, 50)
    assert np.array_equal(num.cpu().numpy(), w_num)
    assert w_num.min() > 5                       # the case is not degenerate
    got = dets.cpu().numpy()
    assert np.array_equal(got[..., 5], w_dets[..., 5])                           # classes, order
    assert np.array_equal(got.view(np.uint32), w_dets.view(np.uint32))           # boxes and scores, bit for bit


def test_detection_postprocess_empty_and_errors(hip):
    import torch
    from mxdetection_amd.core.evaluation import DetectionPostprocess
    N, R, C = 1, 64, 5
    fused = torch.zeros((N * R, 64), device="cuda")            # uniform logits: score 0.2 each
    rois = torch.zeros((N * R, 5), device="cuda")
    rois[:, 3:] = 31.0
    info = torch.tensor([[64.0, 64.0, 1.0]], device="cuda")
    post = DetectionPostprocess(C, score_thresh=0.5, nms_thresh=0.5, max_per_image=10)
    dets, num = post(fused[:, :C], fused[:, C:], rois, torch.tensor([64], dtype=torch.int32, device="cuda"), info)
    assert int(num[0]) == 0 and float(dets[0, :, 5].max()) == -1.0   # nothing above the threshold
    post = DetectionPostprocess(C, score_thresh=0.1, nms_thresh=0.5, max_per_image=10)
    dets, num = post(fused[:, :C], fused[:, C:], rois, torch.tensor([0], dtype=torch.int32, device="cuda"), info)
    assert int(num[0]) == 0                                          # no valid rois
    dets, num = post(fused[:, :C], fused[:, C:], rois, torch.tensor([64], dtype=torch.int32, device="cuda"), info)
    assert int(num[0]) == C - 1                                      # identical boxes: one survivor per class
    with pytest.raises(hip.MxdetError):
        DetectionPostprocess(C, max_per_image=1000)(fused[:, :C], fused[:, C:], rois,
                                                    torch.tensor([64], dtype=torch.int32, device="cuda"), info)


def test_faster_rcnn_predict_runs(hip):
    import torch
    from mxdetection_amd.models import FasterRCNN
    m = FasterRCNN("cuda", seed=7, pre_nms_top_n=600, post_nms_top_n=300)
    torch.manual_seed(0)
    img = torch.randn(2, 3, 192, 256).cuda()
    info = torch.tensor([[192.0, 256.0, 1.0]] * 2).cuda()
    dets, num = m.predict(img, info, score_thresh=0.0, max_per_image=20)
    torch.cuda.synchronize()
    d, k = dets.cpu().numpy(), num.cpu().numpy()
    assert d.shape == (2, 20, 6) and np.all(np.isfinite(d)) and np.all(k == 20)
    for n in range(2):
        assert np.all(d[n, :, 5] >= 1) and np.all(np.diff(d[n, :, 4]) <= 0)          # foreground classes, sorted by score
        assert np.all(d[n, :, 0] >= 0) and np.all(d[n, :, 2] <= 255) and np.all(d[n, :, 3] <= 191)


def test_detection_postprocess_matches_committed_golden(hip):
    """The HIP path against the committed fixture (tests/golden/det_small.npz, pp_*), bit for bit."""
    import os
    import torch
    from mxdetection_amd.core.evaluation import DetectionPostprocess
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_small.npz"))
    Cn = g["pp_cls"].shape[1]
    post = DetectionPostprocess(Cn, score_thresh=0.05, nms_thresh=0.5, max_per_image=20, stds=(0.1, 0.1, 0.2, 0.2))
    dets, num = post(torch.from_numpy(g["pp_cls"]).cuda(), torch.from_numpy(g["pp_reg"]).cuda(),
                     torch.from_numpy(g["pp_rois"]).cuda(), torch.tensor([40, 33], dtype=torch.int32).cuda(),
                     torch.from_numpy(g["prop_info"]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(num.cpu().numpy(), g["pp_num"])
    assert np.array_equal(dets.cpu().numpy().view(np.uint32), g["pp_dets"].view(np.uint32))
