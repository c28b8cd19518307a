"""core/mask (/root/reference/README.md:18): mask-target generation and the mask loss for Mask R-CNN."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def mask_target(rois, matched_gt, labels, gt_masks, size=28, out=None):
    """rois [R,5] f32, matched_gt/labels [R] i32, gt_masks [N,G,H,W] u8 -> (targets [R,S,S] u8, cls [R] i32)."""
    lib = _lib.load()
    R = rois.shape[0]
    N, G, H, W = gt_masks.shape
    if out is None:
        tg = torch.empty((R, size, size), dtype=torch.uint8, device=rois.device)
        cls = torch.empty((R,), dtype=torch.int32, device=rois.device)
    else:
        tg, cls = out
    check(lib.mxdet_mask_target(ptr(rois), ptr(matched_gt), ptr(labels), ptr(gt_masks), R, G, H, W, size, ptr(tg),
                                ptr(cls), stream_ptr()), "mask_target")
    return tg, cls


def mask_loss_workspace(R, S, device):
    return torch.empty((_lib.load().mxdet_mask_loss_workspace_bytes(R, S),), dtype=torch.uint8, device=device)


def mask_loss(logits, cls, targets, loss_out, grad, workspace, loss_scale=1.0):
    """logits/grad bf16 [R,S,S,Cpad]; cls [R] i32; targets [R,S,S] u8; loss_out f32[1]."""
    lib = _lib.load()
    R, S, _, Cpad = logits.shape
    check(lib.mxdet_mask_loss(ptr(logits), ptr(cls), ptr(targets), R, S, Cpad, loss_scale, ptr(loss_out), ptr(grad),
                              ptr(workspace), workspace.numel(), stream_ptr()), "mask_loss")
    return loss_out


def mask_paste(logits, dets, H, W, thresh=0.5, out=None, workspace=None):
    """Inference paste-back: logits bf16 [R,S,S,Cpad], dets [R,6] f32 (x1,y1,x2,y2,score,class) in the output frame
    -> full-frame instance masks [R,H,W] u8 (mxdet_mask_paste)."""
    lib = _lib.load()
    R, S, _, Cpad = logits.shape
    need = lib.mxdet_mask_paste_workspace_bytes(R, S)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=logits.device)
    if out is None:
        out = torch.empty((R, H, W), dtype=torch.uint8, device=logits.device)
    check(lib.mxdet_mask_paste(ptr(logits), ptr(dets), R, S, Cpad, H, W, thresh, ptr(out), ptr(workspace), workspace.numel(),
                               stream_ptr()), "mask_paste")
    return out
