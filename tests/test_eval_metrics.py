"""COCO / VOC accuracy metrics (core/evaluation, README.md:20): hand-computed known answers. pycocotools and the VOC
devkit are not in the image -- parity with them is unpinned."""
import numpy as np
import pytest

from mxdetection_amd.core.evaluation import coco_bbox_eval, detections_to_coco, voc_ap, voc_eval


def _gt(im, cat, box, **kw):
    return dict(image_id=im, category_id=cat, bbox=list(box), **kw)


def _dt(im, cat, box, score):
    return dict(image_id=im, category_id=cat, bbox=list(box), score=score)


def test_perfect_and_empty_detections():
    gts = [_gt(1, 1, (10, 10, 50, 40)), _gt(1, 2, (100, 20, 120, 200)), _gt(2, 1, (5, 5, 20, 20))]
    dts = [_dt(g["image_id"], g["category_id"], g["bbox"], 0.9) for g in gts]
    r = coco_bbox_eval(gts, dts)
    assert r["AP"] == pytest.approx(1.0) and r["AP50"] == pytest.approx(1.0) and r["AR100"] == pytest.approx(1.0)
    assert r["APs"] == pytest.approx(1.0) and r["APm"] == pytest.approx(1.0) and r["APl"] == pytest.approx(1.0)
    z = coco_bbox_eval(gts, [])
    assert z["AP"] == 0.0 and z["AR100"] == 0.0


def test_false_positive_ahead_of_true_positive():
    # one object; a higher-scoring miss comes first: precision at every recall level is 1/2
    gts = [_gt(1, 1, (0, 0, 100, 100))]
    dts = [_dt(1, 1, (300, 300, 50, 50), 0.9), _dt(1, 1, (0, 0, 100, 100), 0.8)]
    r = coco_bbox_eval(gts, dts)
    assert r["AP"] == pytest.approx(0.5) and r["AR1"] == 0.0 and r["AR10"] == pytest.approx(1.0)
    # the other way round: precision 1 at every recall level (the trailing miss does not hurt)
    dts[0]["score"] = 0.1
    assert coco_bbox_eval(gts, dts)["AP"] == pytest.approx(1.0)


def test_iou_threshold_sweep():
    # detection shifted so that IoU = 100*80 / (2*100*100 - 100*80) = 2/3: a match at 0.50..0.65 only (4 of 10 thresholds)
    gts = [_gt(1, 1, (0, 0, 100, 100))]
    dts = [_dt(1, 1, (20, 0, 100, 100), 0.9)]
    r = coco_bbox_eval(gts, dts)
    assert r["AP50"] == pytest.approx(1.0) and r["AP75"] == 0.0 and r["AP"] == pytest.approx(0.4)


def test_crowd_and_area_ranges():
    # a detection inside a crowd region is ignored (neither TP nor FP); a duplicate on a normal object is a FP
    gts = [_gt(1, 1, (0, 0, 200, 200), iscrowd=1), _gt(1, 1, (300, 300, 40, 40))]
    dts = [_dt(1, 1, (10, 10, 50, 50), 0.95), _dt(1, 1, (300, 300, 40, 40), 0.9), _dt(1, 1, (301, 300, 40, 40), 0.5)]
    r = coco_bbox_eval(gts, dts)
    assert r["AP"] == pytest.approx(1.0)                 # TP first among the counted detections; the duplicate trails
    assert r["APm"] == pytest.approx(1.0) and r["APs"] == -1.0 and r["APl"] == -1.0       # 40x40 = medium only
    # two recall levels: objects A (found first) and B (found after one miss) -> precision 1 up to recall .5, 2/3 after
    gts = [_gt(1, 1, (0, 0, 50, 50)), _gt(1, 1, (200, 200, 50, 50))]
    dts = [_dt(1, 1, (0, 0, 50, 50), 0.9), _dt(1, 1, (400, 400, 50, 50), 0.8), _dt(1, 1, (200, 200, 50, 50), 0.7)]
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101.0         # recall thresholds 0..0.50 (51 of them) see precision 1
    assert coco_bbox_eval(gts, dts)["AP"] == pytest.approx(want)


def test_voc_metrics():
    assert voc_ap([0.5, 1.0], [1.0, 2.0 / 3.0]) == pytest.approx(0.5 + 0.5 * 2.0 / 3.0)
    assert voc_ap([0.5, 1.0], [1.0, 2.0 / 3.0], use_07_metric=True) == pytest.approx((6 * 1.0 + 5 * 2.0 / 3.0) / 11.0)
    gts = [dict(image_id=1, category_id=1, bbox=[0, 0, 49, 49]), dict(image_id=1, category_id=1, bbox=[200, 200, 249, 249]),
           dict(image_id=1, category_id=2, bbox=[10, 10, 30, 30], difficult=1)]
    dts = [dict(image_id=1, category_id=1, bbox=[0, 0, 49, 49], score=0.9),
           dict(image_id=1, category_id=1, bbox=[400, 400, 449, 449], score=0.8),
           dict(image_id=1, category_id=1, bbox=[200, 200, 249, 249], score=0.7),
           dict(image_id=1, category_id=2, bbox=[10, 10, 30, 30], score=0.9)]
    m, aps = voc_eval(gts, dts)
    assert aps[1] == pytest.approx(0.5 + 0.5 * 2.0 / 3.0) and aps[2] == 0.0      # only a difficult object: no positives
    assert m == pytest.approx(aps[1] / 2)


def test_detections_to_coco_rescales():
    dets = np.zeros((1, 3, 6), np.float32)
    dets[0, 0] = (20, 40, 119, 79, 0.7, 3)
    out = detections_to_coco(dets, np.array([1]), [42], [2.0], class_to_cat={3: 18})
    assert out == [{"image_id": 42, "category_id": 18, "bbox": [10.0, 20.0, 50.5, 20.5], "score": pytest.approx(0.7)}]


def test_segm_eval_known_answers():
    from mxdetection_amd.core.evaluation import coco_segm_eval
    H, W = 64, 64

    def rect(x1, y1, x2, y2):
        m = np.zeros((H, W), np.uint8)
        m[y1:y2, x1:x2] = 1
        return m
    gts = [dict(image_id=1, category_id=1, mask=rect(0, 0, 40, 40)), dict(image_id=1, category_id=2, mask=rect(10, 40, 60, 60))]
    dts = [dict(image_id=1, category_id=1, mask=rect(0, 0, 40, 40), score=0.9), dict(image_id=1, category_id=2, mask=rect(10, 40, 60, 60), score=0.8)]
    assert coco_segm_eval(gts, dts)["AP"] == pytest.approx(1.0)
    # mask IoU 2/3 (40x32 of a 40x40 object, shifted by 8): matches at IoU 0.50..0.65 only, like the box case
    dts[0]["mask"] = rect(8, 0, 48, 40)
    gts1, dts1 = gts[:1], dts[:1]
    r = coco_segm_eval(gts1, dts1)
    assert r["AP50"] == pytest.approx(1.0) and r["AP75"] == 0.0 and r["AP"] == pytest.approx(0.4)
    # same boxes, different shapes: an L-shaped object against its bounding rectangle has mask IoU 0.75 < box IoU 1
    L = rect(0, 0, 40, 40)
    L[0:20, 20:40] = 0
    r = coco_segm_eval([dict(image_id=1, category_id=1, mask=L)], [dict(image_id=1, category_id=1, mask=rect(0, 0, 40, 40), score=1.0)])
    assert r["AP"] == pytest.approx(0.6)            # IoU 0.75: thresholds 0.50..0.75 (6 of 10)


def test_segm_eval_with_missing_detections():
    """A category / image without any detection (or without ground truth) must not break the mask evaluation."""
    from mxdetection_amd.core.evaluation import coco_segm_eval
    m = np.zeros((16, 16), np.uint8)
    m[2:10, 2:10] = 1
    gts = [dict(image_id=1, category_id=1, mask=m), dict(image_id=2, category_id=2, mask=m)]
    dts = [dict(image_id=1, category_id=1, mask=m, score=0.9), dict(image_id=3, category_id=3, mask=m, score=0.5)]
    r = coco_segm_eval(gts, dts)
    assert r["AP"] == pytest.approx(0.5)            # category 1 perfect, category 2 missed, category 3 has no ground truth
    assert coco_segm_eval(gts, [])["AP"] == 0.0
