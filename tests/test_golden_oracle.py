"""The C oracle reproduces the committed golden vectors (tests/golden/det_small.npz, made by make_golden.py)."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "det_small.npz"))
SHAPES, STRIDES = [(12, 16), (6, 8)], [8, 16]


def test_nms_golden(oracle):
    assert np.array_equal(oracle.nms(G["nms_boxes"], 0.5), G["nms_keep_0p5"])
    assert np.array_equal(oracle.nms(G["nms_boxes"], 0.7), G["nms_keep_0p7"])
    assert 0 < len(G["nms_keep_0p5"]) < len(G["nms_keep_0p7"]) < 200


def test_proposal_golden(oracle):
    base = [oracle.base_anchors(s) for s in STRIDES]
    rois, rs, ra, nr = oracle.proposal([G["prop_sc0"], G["prop_sc1"]], [G["prop_dl0"], G["prop_dl1"]], base, [12, 6],
                                       [16, 8], STRIDES, G["prop_info"], 100, 60, 0.7, 2.0)
    assert np.array_equal(rois, G["prop_rois"]) and np.array_equal(ra, G["prop_anchor"]) and np.array_equal(nr, G["prop_num"])
    # size-independent properties: scores sorted descending, boxes inside the image, anchors unique
    for n in range(2):
        k = nr[n]
        assert np.all(np.diff(rs[n, :k]) <= 0)
        assert len(set(ra[n, :k].tolist())) == k
        assert np.all(rois[n, :k, 1] >= 0) and np.all(rois[n, :k, 3] <= G["prop_info"][n, 1] - 1)


def test_anchor_target_golden(oracle):
    lab, _, tg, mi = oracle.anchor_target(G["at_anchors"], G["at_gt"], G["prop_info"], 0.7, 0.3, 0.0, 64, 0.5, 99, 1, 0)
    assert np.array_equal(lab, G["at_labels"]) and np.array_equal(tg, G["at_targets"]) and np.array_equal(mi, G["at_max_iou"])
    assert all((lab[n] >= 0).sum() <= 64 and (lab[n] == 1).sum() <= 32 for n in range(2))
    assert np.all(tg[lab != 1] == 0)


def test_proposal_target_golden(oracle):
    out = oracle.proposal_target(G["pt_rois_in"], np.array([60, 45], np.int32), G["at_gt"], 32, 0.25, 0.5, 0.5, 0.0, 81,
                                 False, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 99, 1, 0)
    assert np.array_equal(out[0], G["pt_rois"]) and np.array_equal(out[1], G["pt_labels"])
    assert np.array_equal(out[2], G["pt_targets"]) and np.array_equal(out[5], G["pt_num_fg"])
    assert np.all(out[5] <= 8)    # at most fg_fraction * rois_per_image foreground


def test_roi_align_golden(oracle):
    out = oracle.roi_align([G["ra_f0"], G["ra_f1"]], [1 / 8, 1 / 16], G["ra_rois"], G["ra_levels"], 7, 7, 2, 3)
    assert np.array_equal(out, G["ra_out"])
    # RoIAlign of a constant map is that constant
    const = [np.full_like(G["ra_f0"], 0x4040), np.full_like(G["ra_f1"], 0x4040)]   # bf16 3.0
    out = oracle.roi_align(const, [1 / 8, 1 / 16], [[0, 10, 10, 60, 50]], [3], 7, 7, 2, 3)
    assert np.all(out == 0x4040)


def test_detection_postprocess_golden(oracle):
    g = G
    dets, num, _, _ = oracle.detection_postprocess(g["pp_cls"], g["pp_reg"], g["pp_rois"], [40, 33], g["prop_info"],
                                                   (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 0.05, 0.5, 20)
    assert np.array_equal(num, g["pp_num"]) and np.array_equal(dets.view(np.uint32), g["pp_dets"].view(np.uint32))
    assert g["pp_num"].min() >= 10
