"""core/anchor (/root/reference/README.md:16): anchor generation and anchor<->GT target assignment."""
import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def generate_base_anchors(base_size=16, ratios=(0.5, 1.0, 2.0), scales=(8,)):
    """py-faster-rcnn `generate_anchors`: [len(ratios)*len(scales), 4] float32, ratio-major."""
    w = h = float(base_size)
    xc, yc = 0.5 * (w - 1.0), 0.5 * (h - 1.0)
    size = w * h
    out = []
    for r in ratios:
        ws = np.round(np.sqrt(size / r))
        hs = np.round(ws * r)
        for s in scales:
            wss, hss = ws * s, hs * s
            out.append([xc - 0.5 * (wss - 1.0), yc - 0.5 * (hss - 1.0), xc + 0.5 * (wss - 1.0), yc + 0.5 * (hss - 1.0)])
    return np.asarray(out, dtype=np.float32)


def generate_anchors(base_anchors, H, W, stride):
    """Dense anchors of one level, [(y*W+x)*A+a, 4] on the GPU."""
    lib = _lib.load()
    assert base_anchors.is_cuda and base_anchors.dtype == torch.float32 and base_anchors.is_contiguous()
    A = base_anchors.shape[0]
    out = torch.empty((H * W * A, 4), dtype=torch.float32, device=base_anchors.device)
    check(lib.mxdet_generate_anchors(ptr(base_anchors), A, H, W, stride, ptr(out), stream_ptr()), "generate_anchors")
    return out


class AnchorTargetWorkspace:
    """Caller-owned scratch for assign_anchor (the C-ABI never allocates)."""

    def __init__(self, N, A_total, G_max, device):
        lib = _lib.load()
        nbytes = lib.mxdet_anchor_target_workspace_bytes(N, A_total, G_max)
        self.buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.nbytes = nbytes


def assign_anchor(anchors, gt_boxes, im_info, fg_thresh=0.7, bg_thresh=0.3, allowed_border=0.0, batch_size=256,
                  fg_fraction=0.5, seed=0, step=0, image_offset=0, workspace=None, out=None, step_dev=None):
    """RPN / RetinaNet targets. anchors [A,4], gt_boxes [N,G,5], im_info [N,3].

    Returns (labels [N,A] i32, matched_gt [N,A] i32, bbox_targets [N,A,4] f32, max_iou [N,A] f32).
    """
    lib = _lib.load()
    N, G = gt_boxes.shape[0], gt_boxes.shape[1]
    A = anchors.shape[0]
    dev = anchors.device
    if workspace is None:
        workspace = AnchorTargetWorkspace(N, A, G, dev)
    if out is None:
        labels = torch.empty((N, A), dtype=torch.int32, device=dev)
        matched = torch.empty((N, A), dtype=torch.int32, device=dev)
        targets = torch.empty((N, A, 4), dtype=torch.float32, device=dev)
        max_iou = torch.empty((N, A), dtype=torch.float32, device=dev)
    else:
        labels, matched, targets, max_iou = out
    check(lib.mxdet_anchor_target(ptr(anchors), A, ptr(gt_boxes), N, G, ptr(im_info), fg_thresh, bg_thresh,
                                  allowed_border, batch_size, fg_fraction, seed, step, ptr(step_dev), image_offset, ptr(labels),
                                  ptr(matched), ptr(targets), ptr(max_iou), ptr(workspace.buf), workspace.nbytes,
                                  stream_ptr()), "anchor_target")
    return labels, matched, targets, max_iou
