"""Generates tests/golden/det_small.npz: seeded inputs and the C oracle's outputs for small detection-op cases.

The reference ships no fixtures (/root/reference = README.md + LICENSE), so these are vectors of THIS repo's oracle
("parity unpinned"); they pin the oracle against regressions and give the GPU tests committed expected outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_boxes, synth_gt  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    d = {}
    # NMS
    b = synth_boxes(rng, 200, 300, 400)
    b[100:] = b[:100] + rng.uniform(-5, 5, (100, 4)).astype(np.float32)
    d["nms_boxes"] = b
    d["nms_keep_0p5"] = O.nms(b, 0.5)
    d["nms_keep_0p7"] = O.nms(b, 0.7)
    # proposal, 2 levels, bf16-valued logits (ties)
    shapes, strides, A, N = [(12, 16), (6, 8)], [8, 16], 3, 2
    sc = [O.round_bf16(rng.standard_normal((N, H * W * A)).astype(np.float32)) for H, W in shapes]
    dl = [O.round_bf16((rng.standard_normal((N, H * W * A, 4)) * 0.3).astype(np.float32)) for H, W in shapes]
    base = [O.base_anchors(s) for s in strides]
    info = np.array([[96, 128, 1.0], [90, 120, 1.0]], np.float32)
    rois, rs, ra, nr = O.proposal(sc, dl, base, [12, 6], [16, 8], strides, info, 100, 60, 0.7, 2.0)
    d.update(prop_sc0=sc[0], prop_sc1=sc[1], prop_dl0=dl[0], prop_dl1=dl[1], prop_info=info, prop_rois=rois,
             prop_scores=rs, prop_anchor=ra, prop_num=nr)
    # anchor target
    anchors = np.concatenate([O.grid_anchors(base[l], H, W, strides[l]) for l, (H, W) in enumerate(shapes)])
    gt = synth_gt(rng, 2, 6, 96, 128, 2, 5)
    lab, mg, tg, mi = O.anchor_target(anchors, gt, info, 0.7, 0.3, 0.0, 64, 0.5, 99, 1, 0)
    d.update(at_anchors=anchors, at_gt=gt, at_labels=lab, at_targets=tg, at_max_iou=mi)
    # proposal target
    pr = np.zeros((2, 60, 5), np.float32)
    pr[:, :, 1:] = np.stack([synth_boxes(rng, 60, 96, 128) for _ in range(2)])
    pr[0, :10, 1:] = gt[0, 0, :4] + rng.uniform(-3, 3, (10, 4))
    pr[1, :, 0] = 1
    out = O.proposal_target(pr, np.array([60, 45], np.int32), gt, 32, 0.25, 0.5, 0.5, 0.0, 81, False, (0, 0, 0, 0),
                            (0.1, 0.1, 0.2, 0.2), 99, 1, 0)
    d.update(pt_rois_in=pr, pt_rois=out[0], pt_labels=out[1], pt_targets=out[2][..., :].astype(np.float32),
             pt_matched=out[4], pt_num_fg=out[5])
    # RoIAlign forward
    feats = [O.f32_to_bf16_bits(rng.standard_normal((2, H, W, 16)).astype(np.float32)) for H, W in shapes]
    rr = np.concatenate([rng.integers(0, 2, (20, 1)).astype(np.float32), synth_boxes(rng, 20, 96, 128)], 1)
    lv = rng.integers(3, 5, 20).astype(np.int32)
    d.update(ra_f0=feats[0], ra_f1=feats[1], ra_rois=rr, ra_levels=lv,
             ra_out=O.roi_align(feats, [1 / 8, 1 / 16], rr, lv, 7, 7, 2, 3))
    # test-time post-processing (core/evaluation): 2 images x 40 rois x 7 classes, bf16-valued head outputs
    R, Cn = 40, 7
    pp_rois = np.zeros((2 * R, 5), np.float32)
    pp_rois[:, 0] = np.repeat(np.arange(2), R)
    pp_rois[:, 1:] = np.concatenate([synth_boxes(rng, R, 96, 128) for _ in range(2)])
    pp_cls = O.round_bf16((rng.standard_normal((2 * R, Cn)) * 2.0).astype(np.float32))
    pp_reg = O.round_bf16((rng.standard_normal((2 * R, 4 * Cn)) * 0.5).astype(np.float32))
    dets, num, _, _ = O.detection_postprocess(pp_cls, pp_reg, pp_rois, [40, 33], info, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2),
                                              0.05, 0.5, 20)
    d.update(pp_rois=pp_rois, pp_cls=pp_cls, pp_reg=pp_reg, pp_dets=dets, pp_num=num)
    # data pipeline (process_data / datasets): two small frames, one flipped, fractional scales; polygons incl. a
    # concave one, a two-polygon instance and an empty instance. Own generator: the arrays above stay untouched.
    r2 = np.random.default_rng(77)
    im0 = r2.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    im1 = r2.integers(0, 256, (19, 11, 3), dtype=np.uint8)
    dp_scales = np.array([1.37, 2.0 / 3.0], np.float64)
    dp_out, dp_u8 = O.image_preprocess([im0, im1], dp_scales.tolist(), [False, True], 32, 32, (123.68, 116.779, 103.939),
                                       (58.4, 57.1, 57.4), swap_rb=True, return_u8=True)
    polys = [[[2.0, 1.5, 20.5, 3.0, 18.0, 14.2, 9.0, 6.0, 3.5, 15.0]],                  # concave
             [[1.0, 1.0, 6.0, 1.0, 6.0, 6.0, 1.0, 6.0], [4.0, 4.0, 12.0, 5.0, 8.0, 11.0]],   # union of two
             []]
    from mxdetection_amd.process_data.transform import pack_polygons
    pv, ps, pf = pack_polygons([[[np.asarray(q, np.float32).reshape(-1, 2) for q in inst] for inst in polys]], 1, 3)
    d.update(dp_im0=im0, dp_im1=im1, dp_scales=dp_scales, dp_out=dp_out, dp_u8_0=dp_u8[0], dp_u8_1=dp_u8[1],
             pm_verts=pv, pm_start=ps, pm_first=pf, pm_masks=O.polygon_masks(pv, ps, pf, 3, 16, 24))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "det_small.npz"), **d)
    print("wrote det_small.npz with", len(d), "arrays")


if __name__ == "__main__":
    main()
