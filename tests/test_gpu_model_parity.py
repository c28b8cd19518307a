"""Whole-path parity: the HIP Faster R-CNN step vs the CPU restatement (oracle/model_ref.py) with the SAME weights.

Discrete decisions are compared exactly (proposals / sampled rois / labels, via the C oracle on the HIP head
outputs); floating-point results (head outputs, the four losses, weight gradients) within the tolerance the bf16
pipeline implies, stated at each assert. The oracle is "parity unpinned" by the reference (no reference code).
"""
import numpy as np
import pytest

from conftest import synth_gt

pytestmark = pytest.mark.gpu


def test_step_matches_cpu_restatement(hip, oracle):
    import torch
    from mxdetection_amd.models import FasterRCNN
    from mxdetection_amd.models.rpn_heads.rpn_head import HEAD_CPAD
    from oracle import model_ref as M
    torch.set_num_threads(8)
    N, H, W = 1, 192, 256
    rng = np.random.default_rng(5)
    gt = synth_gt(rng, N, 8, H, W - 6, 2, 6)
    info = np.array([[H, W - 6, 1.0]] * N, np.float32)
    img = torch.randn((N, 3, H, W), generator=torch.Generator().manual_seed(3))
    img = img.to(torch.bfloat16).float()    # both sides see the same bf16-representable pixels
    pre, post, R = 600, 300, 128
    m = FasterRCNN("cuda", seed=7, pre_nms_top_n=pre, post_nms_top_n=post, rois_per_image=R)
    rpn_l, rcnn_l = m.forward_backward(img.cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(info).cuda(), step=2,
                                       image_offset=5)
    torch.cuda.synchronize()
    got_losses = torch.cat([rpn_l, rcnn_l]).cpu().numpy()
    ref = M.RefModel(m.export_params(), pre_n=pre, post_n=post, rois_per_image=R)

    # (1) discrete path, exact: oracle proposal + proposal-target on the HIP head outputs == HIP rois / labels
    A = 3
    hs = [h.float().cpu().numpy() for h in m.rpn_head.h]
    shapes = [(h.shape[1], h.shape[2]) for h in hs]
    sc = [h[..., :A].reshape(N, -1) for h in hs]
    dl = [h[..., A:5 * A].reshape(N, -1, 4) for h in hs]
    base = [oracle.base_anchors(s) for s in M.STRIDES]
    rois, _, _, num = oracle.proposal(sc, dl, base, [s[0] for s in shapes], [s[1] for s in shapes], M.STRIDES, info, pre,
                                      post, 0.7, 0.0)
    srois, slab, stgt, swgt, _, nfg = oracle.proposal_target(rois, num, gt, R, 0.25, 0.5, 0.5, 0.0, 81, False, (0, 0, 0, 0),
                                                             (0.1, 0.1, 0.2, 0.2), 99, 2, 5)
    assert np.array_equal(m.bbox_head.rois.cpu().numpy().view(np.uint32), srois.view(np.uint32))
    assert np.array_equal(m.bbox_head.labels.cpu().numpy(), slab)
    assert np.array_equal(m.bbox_head.num_fg.cpu().numpy(), nfg)

    # (2) floating point, teacher-forced with the HIP proposals so both sides sample the same rois
    out = ref.step(img, gt, info, step=2, image_offset=5, forced_rois=(rois, num))
    assert np.array_equal(out["sampled_rois"], srois)
    # head outputs: ~50 bf16 layers deep; tolerance 3% of the tensor's rms + 3% relative
    for l, h in enumerate(hs):
        r = out["heads"][l].detach().permute(0, 2, 3, 1).numpy()[..., :5 * A]
        g = h[..., :5 * A]
        rms = np.sqrt((r ** 2).mean())
        assert np.abs(g - r).max() <= 0.06 * rms + 0.03 * np.abs(r).max(), (l, np.abs(g - r).max(), rms)
        assert np.all(h[..., 5 * A:HEAD_CPAD] == 0)   # padding channels: zero filters, zero bias, zero gradient
    # losses: within 0.5% (bf16 activations; measured 3e-4 .. 2e-3) of the fp32 CPU restatement
    assert np.allclose(got_losses, out["losses"], rtol=5e-3, atol=1e-3), (got_losses, out["losses"])
    # gradients: relative L2 error per parameter tensor. Activations AND gradients are stored in bf16 (2^-9 relative
    # rounding per tensor), so the error grows with depth. Bounds = twice the values measured in round 2 (the kernels'
    # own arithmetic is checked layer by layer at 1e-3 in tests/test_gpu_dense.py::test_layer_gradients_vs_torch_fp32;
    # what accumulates here is the bf16 storage of ~50 tensors between the loss and layer2).
    grads = m.export_grads()
    bounds = {"bbox.fc_out.weight": 3e-3, "bbox.fc1.weight": 2e-2, "rpn.out.weight": 5e-3, "rpn.conv.weight": 3.2e-2,
              "fpn.out2.weight": 1.8e-2, "fpn.lat5.weight": 2e-2, "fpn.lat2.weight": 1.8e-2, "layer4.2.conv3.weight": 3.2e-2,
              "layer4.0.down.weight": 4.5e-2, "layer3.0.conv2.weight": 6.6e-2, "layer2.0.conv1.weight": 8e-2,
              "bbox.fc2.bias": 1.3e-2, "fpn.out3.bias": 1.2e-1, "rpn.conv.bias": 3.3e-2}
    worst = 0.0
    for name, bound in bounds.items():
        g, r = grads[name].numpy().ravel(), out["grads"][name].numpy().ravel()
        rel = np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-30)
        worst = max(worst, rel)
        print("grad rel-L2", name, float(rel))
        assert rel < bound, (name, rel, bound)
    print("losses hip", got_losses, "ref", out["losses"], "worst grad rel-L2", worst)
