"""core/loss (/root/reference/README.md:19): smooth-L1, RPN / box-head losses, sigmoid focal loss."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


def smooth_l1(pred, target, weight=None, sigma=1.0):
    lib = _lib.load()
    out = torch.empty_like(pred)
    check(lib.mxdet_smooth_l1_fwd(ptr(pred), ptr(target), ptr(weight), pred.numel(), sigma, ptr(out), stream_ptr()),
          "smooth_l1_fwd")
    return out


def smooth_l1_backward(pred, target, weight=None, grad_out=None, sigma=1.0, grad_pred=None, accumulate=False):
    lib = _lib.load()
    if grad_pred is None:
        grad_pred = torch.empty_like(pred)
    check(lib.mxdet_smooth_l1_bwd(ptr(pred), ptr(target), ptr(weight), ptr(grad_out), pred.numel(), sigma,
                                  int(accumulate), ptr(grad_pred), stream_ptr()), "smooth_l1_bwd")
    return grad_pred


def loss_workspace(n, device):
    lib = _lib.load()
    return torch.empty((lib.mxdet_loss_workspace_bytes(n),), dtype=torch.uint8, device=device)


def focal_loss(logits, labels, alpha=0.25, gamma=2.0, grad_scale=1.0, workspace=None):
    """Sigmoid focal loss fwd+bwd. logits [n,C] (f32|bf16), labels [n] i32. Returns (loss[1] f32, grad)."""
    lib = _lib.load()
    n, Cc = logits.shape
    if workspace is None:
        workspace = loss_workspace(n, logits.device)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits)
    check(lib.mxdet_focal_loss(ptr(logits), _DT[logits.dtype], ptr(labels), n, Cc, alpha, gamma, grad_scale,
                               ptr(loss), ptr(grad), ptr(workspace), workspace.numel(), stream_ptr()), "focal_loss")
    return loss, grad


def rpn_loss_level(head, A, labels, bbox_targets, level_offset, sigma, norm, loss_scale, grad_head, partial):
    """One pyramid level of the RPN loss; head/grad_head bf16 [N,H,W,Cpad]; partial: f32 view for this level."""
    lib = _lib.load()
    N, H, W, Cpad = head.shape
    check(lib.mxdet_rpn_loss_level(ptr(head), N, H, W, A, Cpad, ptr(labels), ptr(bbox_targets), labels.shape[1],
                                   level_offset, sigma, norm, loss_scale, ptr(grad_head), ptr(partial),
                                   stream_ptr()), "rpn_loss_level")


def rpn_loss_num_partials(N, H, W):
    return _lib.load().mxdet_rpn_loss_num_partials(N, H, W)


def loss_finalize(partial, count, ncomp, out):
    check(_lib.load().mxdet_loss_finalize(ptr(partial), count, ncomp, ptr(out), stream_ptr()), "loss_finalize")


def rcnn_loss(cls_logits, bbox_pred, labels, bbox_targets, bbox_weights, num_classes, reg_dim, ld_cls, ld_reg,
              sigma, norm, loss_scale, grad_cls, grad_reg, loss_out, workspace):
    """Box-head losses fwd+bwd over R rois; tensors may be column views of one fused head output."""
    lib = _lib.load()
    R = labels.numel()
    check(lib.mxdet_rcnn_loss(ptr(cls_logits), ptr(bbox_pred), _DT[cls_logits.dtype], ld_cls, ld_reg, ptr(labels),
                              ptr(bbox_targets), ptr(bbox_weights), R, num_classes, reg_dim, sigma, norm, loss_scale,
                              ptr(loss_out), ptr(grad_cls), ptr(grad_reg), ptr(workspace), workspace.numel(),
                              stream_ptr()), "rcnn_loss")


def anchor_class_labels(labels, matched_gt, gt_boxes, cls_labels, num_fg):
    """RetinaNet: labels/matched [N,A] i32 -> cls_labels [N,A] (class id for fg), num_fg i32[1]."""
    lib = _lib.load()
    N, A = labels.shape
    check(lib.mxdet_anchor_class_labels(ptr(labels), ptr(matched_gt), ptr(gt_boxes), N, A, gt_boxes.shape[1],
                                        ptr(cls_labels), ptr(num_fg), stream_ptr()), "anchor_class_labels")


def retina_loss_num_partials(N, H, W, A):
    return _lib.load().mxdet_retina_loss_num_partials(N, H, W, A)


def retina_loss_level(cls, reg, A, C, cls_labels, bbox_targets, level_offset, alpha, gamma, sigma, num_fg, loss_scale,
                      grad_cls, grad_reg, partial):
    lib = _lib.load()
    N, H, W, ld_cls = cls.shape
    check(lib.mxdet_retina_loss_level(ptr(cls), ptr(reg), N, H, W, A, C, ld_cls, reg.shape[3], ptr(cls_labels),
                                      ptr(bbox_targets), cls_labels.shape[1], level_offset, alpha, gamma, sigma,
                                      ptr(num_fg), loss_scale, ptr(grad_cls), ptr(grad_reg), ptr(partial), stream_ptr()),
          "retina_loss_level")
