"""core/evaluation (/root/reference/README.md:20): test-time detection post-processing, on the GPU.

SURVEY.md section 8f rank 3 ("the step immediately after the path"). The MXNet lineage runs `im_detect` ->
per-class score threshold -> nms -> max_per_image in numpy on the host (README.md:37,41-44); here it is one C-ABI call
(`mxdet_detection_postprocess`, csrc/postprocess.hip) that reuses the training path's batched NMS.
"""
import ctypes as C

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


class DetectionPostprocess:
    """softmax + class-specific decode + per-class NMS + top-k per image for a two-stage box head.

    cls_logits [N*R, >=C] and bbox_pred [N*R, >=4C] may be column views of one fused head output (leading dimensions are
    taken from the strides). Returns (dets [N, max_per_image, 6] f32 = x1,y1,x2,y2,score,class; num_dets [N] i32).
    """

    def __init__(self, num_classes=81, score_thresh=0.05, nms_thresh=0.5, max_per_image=100,
                 means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2)):
        self.C, self.score_thresh, self.nms_thresh, self.max_det = num_classes, score_thresh, nms_thresh, max_per_image
        self.means = (C.c_float * 4)(*means)
        self.stds = (C.c_float * 4)(*stds)
        self._ws = None
        self._out = None

    def __call__(self, cls_logits, bbox_pred, rois, num_rois, im_info):
        lib = _lib.load()
        N = im_info.shape[0]
        R = rois.shape[0] // N
        dev = rois.device
        need = lib.mxdet_detection_postprocess_workspace_bytes(N, R, self.C)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        if self._out is None or self._out[0].shape[0] != N:
            self._out = (torch.empty((N, self.max_det, 6), dtype=torch.float32, device=dev),
                         torch.empty((N,), dtype=torch.int32, device=dev))
        dets, num = self._out
        check(lib.mxdet_detection_postprocess(ptr(cls_logits), ptr(bbox_pred), _DT[cls_logits.dtype],
                                              cls_logits.stride(0), bbox_pred.stride(0), ptr(rois), ptr(num_rois),
                                              ptr(im_info), N, R, self.C, self.means, self.stds, self.score_thresh,
                                              self.nms_thresh, self.max_det, ptr(dets), ptr(num), ptr(self._ws),
                                              self._ws.numel(), stream_ptr()), "detection_postprocess")
        return dets, num


# ---------------------------------------------------------------------------------------------------------------------
# Offline accuracy metrics (README.md:20 `core/evaluation`: COCO / VOC mAP). Host-side numpy: they run once per epoch
# on a few thousand boxes, far from the img/s path. Restated from the published COCO evaluation protocol
# (pycocotools.cocoeval: per-image greedy matching in score order at IoU 0.50:0.05:0.95, crowd / out-of-range ground
# truth ignored, 101-point interpolated precision) and the VOC devkit's AP; neither package is in this image, so the
# numbers are pinned by hand-computed cases only (tests/test_eval_metrics.py) -- parity with pycocotools is UNPINNED.
# ---------------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

COCO_IOU_THRS = np.linspace(0.5, 0.95, 10)
COCO_REC_THRS = np.linspace(0.0, 1.0, 101)
COCO_AREAS = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}


def _iou_xywh(dt, gt, crowd):
    """dt [D,4], gt [G,4] as (x, y, w, h), continuous extents; for crowd ground truth the union is the detection area."""
    if dt.shape[0] == 0 or gt.shape[0] == 0:
        return np.zeros((dt.shape[0], gt.shape[0]))
    ix = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    iy = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    da, ga = (dt[:, 2] * dt[:, 3])[:, None], (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(crowd, bool)[None, :], da, da + ga - inter)
    return inter / np.maximum(union, 1e-12)


def _match_image(dt, gt_ig, crowd, ious, dt_area, area_rng):
    """One image, one category, one area range. dt sorted by score desc; gts sorted non-ignored first.
    Returns (matched [T,D] bool, dt_ignore [T,D] bool)."""
    T, D, Gn = len(COCO_IOU_THRS), dt.shape[0], gt_ig.shape[0]
    dtm = np.zeros((T, D), bool)
    dtig = np.zeros((T, D), bool)
    for ti, thr in enumerate(COCO_IOU_THRS):
        gtm = np.zeros((Gn,), bool)
        for d in range(D):
            best, m = min(thr, 1 - 1e-10), -1
            for g in range(Gn):
                if gtm[g] and not crowd[g]:
                    continue
                if m > -1 and not gt_ig[m] and gt_ig[g]:
                    break                                   # a real match exists: do not trade it for an ignored one
                if ious[d, g] < best:
                    continue
                best, m = ious[d, g], g
            if m == -1:
                continue
            dtm[ti, d], dtig[ti, d], gtm[m] = True, gt_ig[m], True
    out = (dt_area < area_rng[0]) | (dt_area > area_rng[1])
    dtig |= (~dtm) & out[None, :]
    return dtm, dtig


def _iou_masks(dm, gm, crowd):
    """dm [D,H,W], gm [G,H,W] binary masks in one frame; for crowd ground truth the union is the detection area."""
    if dm.shape[0] == 0 or gm.shape[0] == 0:
        return np.zeros((dm.shape[0], gm.shape[0]))
    d = dm.reshape(dm.shape[0], -1).astype(np.float32)
    g = gm.reshape(gm.shape[0], -1).astype(np.float32)
    inter = (d @ g.T).astype(np.float64)
    da, ga = d.sum(1).astype(np.float64)[:, None], g.sum(1).astype(np.float64)[None, :]
    union = np.where(np.asarray(crowd, bool)[None, :], da, da + ga - inter)
    return inter / np.maximum(union, 1e-12)


def coco_segm_eval(gts, dts, max_dets=(1, 10, 100)):
    """COCO mask AP / AR: same protocol as coco_bbox_eval with mask IoU. Every entry carries "mask" (binary [H,W] array in
    the image's frame); areas are mask areas unless the ground truth gives "area"."""
    return coco_bbox_eval(gts, dts, max_dets, _segm=True)


def coco_bbox_eval(gts, dts, max_dets=(1, 10, 100), _segm=False):
    """COCO box AP / AR.

    gts: list of dicts {image_id, category_id, bbox [x,y,w,h], iscrowd (0/1, optional), area (optional)};
    dts: list of dicts {image_id, category_id, bbox [x,y,w,h], score}.
    Returns {"AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl"} (-1 where undefined).
    """
    cats = sorted({g["category_id"] for g in gts} | {d["category_id"] for d in dts})
    imgs = sorted({g["image_id"] for g in gts} | {d["image_id"] for d in dts})
    G, Dd = {}, {}
    for g in gts:
        G.setdefault((g["image_id"], g["category_id"]), []).append(g)
    for d in dts:
        Dd.setdefault((d["image_id"], d["category_id"]), []).append(d)
    T, R, K, A, M = len(COCO_IOU_THRS), len(COCO_REC_THRS), len(cats), len(COCO_AREAS), len(max_dets)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for ki, c in enumerate(cats):
        for ai, rng in enumerate(COCO_AREAS.values()):
            per_img = []
            for im in imgs:
                g, d = G.get((im, c), []), Dd.get((im, c), [])
                if not g and not d:
                    continue
                if _segm:
                    shp = (g[0] if g else d[0])["mask"].shape
                    gb = np.array([np.asarray(x["mask"]) > 0 for x in g], bool).reshape((-1,) + shp)
                    ga = np.array([x.get("area", float((np.asarray(x["mask"]) > 0).sum())) for x in g], np.float64)
                else:
                    gb = np.array([x["bbox"] for x in g], np.float64).reshape(-1, 4)
                    ga = np.array([x.get("area", x["bbox"][2] * x["bbox"][3]) for x in g], np.float64)
                crowd = np.array([bool(x.get("iscrowd", 0)) for x in g], bool)
                ig = crowd | (ga < rng[0]) | (ga > rng[1])
                go = np.argsort(ig, kind="mergesort")
                gb, crowd, ig = gb[go], crowd[go], ig[go]
                do = np.argsort([-x["score"] for x in d], kind="mergesort")[:max_dets[-1]]
                sc = np.array([d[i]["score"] for i in do], np.float64)
                if _segm:
                    db = np.array([np.asarray(d[i]["mask"]) > 0 for i in do], bool).reshape((-1,) + shp)
                    ious = _iou_masks(db, gb, crowd)
                    darea = db.reshape(db.shape[0], -1).sum(1).astype(np.float64) if db.shape[0] else np.zeros((0,))
                else:
                    db = np.array([d[i]["bbox"] for i in do], np.float64).reshape(-1, 4)
                    ious, darea = _iou_xywh(db, gb, crowd), db[:, 2] * db[:, 3]
                dtm, dtig = _match_image(db, ig, crowd, ious, darea, rng)
                per_img.append((sc, dtm, dtig, int((~ig).sum())))
            if not per_img:
                continue
            npig = sum(p[3] for p in per_img)
            if npig == 0:
                continue
            for mi, md in enumerate(max_dets):
                sc = np.concatenate([p[0][:md] for p in per_img])
                order = np.argsort(-sc, kind="mergesort")
                dtm = np.concatenate([p[1][:, :md] for p in per_img], axis=1)[:, order]
                dtig = np.concatenate([p[2][:, :md] for p in per_img], axis=1)[:, order]
                tps = np.cumsum(dtm & ~dtig, axis=1).astype(np.float64)
                fps = np.cumsum(~dtm & ~dtig, axis=1).astype(np.float64)
                for ti in range(T):
                    tp, fp = tps[ti], fps[ti]
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[ti, ki, ai, mi] = rc[-1] if rc.size else 0.0
                    for i in range(pr.size - 1, 0, -1):               # precision envelope
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros((R,))
                    idx = np.searchsorted(rc, COCO_REC_THRS, side="left")
                    ok = idx < pr.size
                    q[ok] = pr[idx[ok]]
                    precision[ti, :, ki, ai, mi] = q

    def _ap(ai, ti=None):
        p = precision[:, :, :, ai, M - 1] if ti is None else precision[ti, :, :, ai, M - 1]
        p = p[p > -1]
        return float(p.mean()) if p.size else -1.0

    def _ar(ai, mi):
        r = recall[:, :, ai, mi]
        r = r[r > -1]
        return float(r.mean()) if r.size else -1.0

    names = list(COCO_AREAS)
    out = {"AP": _ap(0), "AP50": _ap(0, 0), "AP75": _ap(0, 5), "APs": _ap(names.index("small")),
           "APm": _ap(names.index("medium")), "APl": _ap(names.index("large"))}
    for mi, md in enumerate(max_dets):
        out["AR%d" % md] = _ar(0, mi)
    out.update(ARs=_ar(1, M - 1), ARm=_ar(2, M - 1), ARl=_ar(3, M - 1))
    return out


def voc_ap(rec, prec, use_07_metric=False):
    """VOC devkit average precision: 11-point (2007) or area under the monotone precision envelope."""
    rec, prec = np.asarray(rec, np.float64), np.asarray(prec, np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            ap += (prec[rec >= t].max() if np.any(rec >= t) else 0.0) / 11.0
        return float(ap)
    mrec = np.concatenate([[0.0], rec, [1.0]])
    mpre = np.concatenate([[0.0], prec, [0.0]])
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def voc_eval(gts, dts, iou_thresh=0.5, use_07_metric=False):
    """VOC mAP over classes. gts: {image_id, category_id, bbox [x1,y1,x2,y2] inclusive pixels, difficult (optional)};
    dts: {image_id, category_id, bbox [x1,y1,x2,y2], score}. Returns (mAP, {category: AP})."""
    aps = {}
    for c in sorted({g["category_id"] for g in gts}):
        gt_by_img = {}
        for g in gts:
            if g["category_id"] == c:
                gt_by_img.setdefault(g["image_id"], []).append(g)
        npos = sum(1 for gl in gt_by_img.values() for g in gl if not g.get("difficult", 0))
        used = {im: np.zeros(len(gl), bool) for im, gl in gt_by_img.items()}
        dl = sorted([d for d in dts if d["category_id"] == c], key=lambda d: -d["score"])
        tp, fp = np.zeros(len(dl)), np.zeros(len(dl))
        for i, d in enumerate(dl):
            gl = gt_by_img.get(d["image_id"], [])
            best, bj = -1.0, -1
            bb = d["bbox"]
            for j, g in enumerate(gl):
                gb = g["bbox"]
                iw = min(bb[2], gb[2]) - max(bb[0], gb[0]) + 1.0
                ih = min(bb[3], gb[3]) - max(bb[1], gb[1]) + 1.0
                if iw > 0 and ih > 0:
                    ua = (bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (gb[2] - gb[0] + 1.0) * (gb[3] - gb[1] + 1.0) - iw * ih
                    if iw * ih / ua > best:
                        best, bj = iw * ih / ua, j
            if best > iou_thresh:
                if gl[bj].get("difficult", 0):
                    continue
                if not used[d["image_id"]][bj]:
                    tp[i], used[d["image_id"]][bj] = 1.0, True
                else:
                    fp[i] = 1.0
            else:
                fp[i] = 1.0
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        rec = ctp / max(npos, 1)
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        aps[c] = voc_ap(rec, prec, use_07_metric) if npos else 0.0
    return (float(np.mean(list(aps.values()))) if aps else 0.0), aps


def detections_to_coco(dets, num_dets, image_ids, scales, class_to_cat=None):
    """Device-side post-processing output -> COCO result dicts in ORIGINAL image coordinates.
    dets [N,M,6] = (x1,y1,x2,y2,score,class) at network-input scale (inclusive corners), scales [N] = resize factors."""
    dets = dets.detach().cpu().numpy() if hasattr(dets, "detach") else np.asarray(dets)
    num = num_dets.detach().cpu().numpy() if hasattr(num_dets, "detach") else np.asarray(num_dets)
    out = []
    for n in range(dets.shape[0]):
        for k in range(int(num[n])):
            x1, y1, x2, y2, s, c = [float(v) for v in dets[n, k]]
            x1, y1, x2, y2 = x1 / scales[n], y1 / scales[n], x2 / scales[n], y2 / scales[n]
            cat = int(c) if class_to_cat is None else class_to_cat[int(c)]
            out.append({"image_id": image_ids[n], "category_id": cat, "bbox": [x1, y1, x2 - x1 + 1.0, y2 - y1 + 1.0], "score": s})
    return out


class RetinaDetect:
    """One-stage test-time detection on the device (mxdet_retina_detect): per-level top-k over (anchor, class) logits,
    decode, merge, sigmoid, per-class NMS, max_per_image. cls[l] bf16/f32 [N,H,W,>=A*C] (channel a*C + c), reg[l]
    [N,H,W,>=4A] (channel a*4 + k), base[l] device f32 [A,4]."""

    def __init__(self, num_classes, strides, base_anchors, pre_nms_top_n=1000, score_thresh=0.05, nms_thresh=0.5,
                 max_per_image=100):
        self.C, self.strides, self.base = num_classes, list(strides), base_anchors
        self.pre_n, self.score_thresh, self.nms_thresh, self.max_det = pre_nms_top_n, score_thresh, nms_thresh, max_per_image
        self._ws = None

    def __call__(self, cls, reg, im_info):
        from .._lib import PyramidT
        lib = _lib.load()
        N = cls[0].shape[0]
        A = self.base[0].shape[0]
        d = PyramidT()
        d.num_levels, d.A, d.classes = len(cls), A * self.C, self.C
        d.dtype = _DT[cls[0].dtype]
        for l, (c, r) in enumerate(zip(cls, reg)):
            assert c.dtype == reg[l].dtype == cls[0].dtype and c.stride(3) == 1 and r.stride(3) == 1
            d.H[l], d.W[l], d.stride[l] = c.shape[1], c.shape[2], self.strides[l]
            d.cls[l], d.reg[l], d.base_anchors[l] = c.data_ptr(), r.data_ptr(), self.base[l].data_ptr()
            d.cls_sn[l], d.cls_sy[l], d.cls_sx[l], d.cls_sa[l] = c.stride(0), c.stride(1), c.stride(2), 1
            d.reg_sn[l], d.reg_sy[l], d.reg_sx[l], d.reg_sc[l] = r.stride(0), r.stride(1), r.stride(2), 1
        need = lib.mxdet_retina_detect_workspace_bytes(C.byref(d), N, self.pre_n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=cls[0].device)
        dets = torch.empty((N, self.max_det, 6), dtype=torch.float32, device=cls[0].device)
        num = torch.empty((N,), dtype=torch.int32, device=cls[0].device)
        check(lib.mxdet_retina_detect(C.byref(d), N, ptr(im_info), self.pre_n, self.score_thresh, self.nms_thresh, self.max_det,
                                      ptr(dets), ptr(num), ptr(self._ws), self._ws.numel(), stream_ptr()), "retina_detect")
        return dets, num
