"""core/bbox (/root/reference/README.md:17): box overlaps and proposal-target sampling on the GPU.

Names follow the py-faster-rcnn / mx-rcnn lineage the reference's declared layout comes from
(`bbox_overlaps`, `sample_rois`); every function launches hand-written HIP through the C-ABI.
"""
import ctypes as C

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected contiguous cuda float32"
    return t


def bbox_overlaps(boxes, query_boxes):
    """IoU matrix [N,K] of boxes[N,4] vs query_boxes[K,4] (legacy +1 convention)."""
    lib = _lib.load()
    _f32c(boxes), _f32c(query_boxes)
    out = torch.empty((boxes.shape[0], query_boxes.shape[0]), dtype=torch.float32, device=boxes.device)
    check(lib.mxdet_box_iou(ptr(boxes), boxes.shape[0], ptr(query_boxes), query_boxes.shape[0], ptr(out),
                            stream_ptr()), "box_iou")
    return out


def sample_rois(rois, num_rois, gt_boxes, rois_per_image=512, fg_fraction=0.25, fg_thresh=0.5, bg_thresh_hi=0.5,
                bg_thresh_lo=0.0, num_classes=81, class_agnostic=False, bbox_means=(0.0, 0.0, 0.0, 0.0),
                bbox_stds=(0.1, 0.1, 0.2, 0.2), seed=0, step=0, image_offset=0, step_dev=None):
    """proposal-target. rois [N,S,5], num_rois [N] int32, gt_boxes [N,G,5] (class < 0 = padding).

    Returns (rois [N,R,5], labels [N,R] i32, bbox_targets [N,R,D], bbox_weights [N,R,D], matched_gt [N,R] i32,
    num_fg [N] i32); D = 4 (class agnostic) or 4*num_classes.
    """
    lib = _lib.load()
    _f32c(rois), _f32c(gt_boxes)
    N, S = rois.shape[0], rois.shape[1]
    G = gt_boxes.shape[1]
    R = rois_per_image
    D = 4 if class_agnostic else 4 * num_classes
    dev = rois.device
    out_rois = torch.empty((N, R, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((N, R), dtype=torch.int32, device=dev)
    tgt = torch.empty((N, R, D), dtype=torch.float32, device=dev)
    wgt = torch.empty((N, R, D), dtype=torch.float32, device=dev)
    matched = torch.empty((N, R), dtype=torch.int32, device=dev)
    num_fg = torch.empty((N,), dtype=torch.int32, device=dev)
    means = (C.c_float * 4)(*bbox_means)
    stds = (C.c_float * 4)(*bbox_stds)
    check(lib.mxdet_proposal_target(ptr(rois), ptr(num_rois), S, ptr(gt_boxes), N, G, R, fg_fraction, fg_thresh,
                                    bg_thresh_hi, bg_thresh_lo, num_classes, int(class_agnostic), means, stds,
                                    seed, step, ptr(step_dev), image_offset, ptr(out_rois), ptr(labels), ptr(tgt), ptr(wgt),
                                    ptr(matched), ptr(num_fg), stream_ptr()), "proposal_target")
    return out_rois, labels, tgt, wgt, matched, num_fg
