"""ops: FPN level map + multi-level RoIAlign (MXNet role: contrib.ROIAlign)."""
import ctypes as C

import torch

from .. import _lib
from .._lib import FeatPyramidT, check, ptr, stream_ptr


def fpn_level_map(rois, lvl_min=2, lvl_max=5):
    lib = _lib.load()
    R = rois.shape[0]
    levels = torch.empty((R,), dtype=torch.int32, device=rois.device)
    check(lib.mxdet_fpn_level_map(ptr(rois), R, lvl_min, lvl_max, ptr(levels), stream_ptr()), "fpn_level_map")
    return levels


def _pyr(feats, scales, lvl_min):
    d = FeatPyramidT()
    d.num_levels, d.lvl_min = len(feats), lvl_min
    for l, f in enumerate(feats):
        d.H[l], d.W[l] = f.shape[1], f.shape[2]
        d.spatial_scale[l] = scales[l]
        d.feat[l] = f.data_ptr()
    return d


def roi_align_forward(feats, scales, rois, levels, pooled=(7, 7), sampling_ratio=2, lvl_min=2, out=None):
    """feats[l]: bf16 [N,H,W,C]; rois [R,5] f32; levels [R] i32 -> bf16 [R,PH,PW,C]."""
    lib = _lib.load()
    N, C_ = feats[0].shape[0], feats[0].shape[3]
    R = rois.shape[0]
    if out is None:
        out = torch.empty((R, pooled[0], pooled[1], C_), dtype=torch.bfloat16, device=rois.device)
    d = _pyr(feats, scales, lvl_min)
    check(lib.mxdet_roi_align_fwd(C.byref(d), N, C_, ptr(rois), ptr(levels), R, pooled[0], pooled[1], sampling_ratio,
                                  ptr(out), stream_ptr()), "roi_align_fwd")
    return out


def roi_align_backward(dfeats, scales, rois, levels, grad_out, sampling_ratio=2, lvl_min=2):
    """Scatter-adds grad_out (bf16 [R,PH,PW,C]) into the fp32 accumulators dfeats[l] ([N,H,W,C])."""
    lib = _lib.load()
    N, C_ = dfeats[0].shape[0], dfeats[0].shape[3]
    R, PH, PW = grad_out.shape[0], grad_out.shape[1], grad_out.shape[2]
    d = _pyr(dfeats, scales, lvl_min)
    check(lib.mxdet_roi_align_bwd(C.byref(d), N, C_, ptr(rois), ptr(levels), R, PH, PW, sampling_ratio,
                                  ptr(grad_out), stream_ptr()), "roi_align_bwd")


_gather_ws = {}


def _gather_workspace(lib, d, N, R, device, workspace):
    need = lib.mxdet_roi_align_bwd_gather_workspace_bytes(C.byref(d), N, R)
    if workspace is None:
        key = (device, need)
        workspace = _gather_ws.get(key)
        if workspace is None:
            workspace = torch.empty((max(need, 256),), dtype=torch.uint8, device=device)
            _gather_ws[key] = workspace
    return workspace


def roi_align_backward_gather_workspace(dmaps, scales, R, lvl_min=2):
    """A workspace of the caller's own (needed when the records are prepared ahead of the gather)."""
    lib = _lib.load()
    d = _pyr(dmaps, scales, lvl_min)
    need = lib.mxdet_roi_align_bwd_gather_workspace_bytes(C.byref(d), dmaps[0].shape[0], R)
    return torch.empty((max(need, 256),), dtype=torch.uint8, device=dmaps[0].device)


def roi_align_backward_gather_prepare(dmaps, scales, rois, levels, pooled, sampling_ratio, lvl_min, workspace):
    """Per-roi records of the gather (they depend only on the rois): issue as soon as the rois exist."""
    lib = _lib.load()
    d = _pyr(dmaps, scales, lvl_min)
    check(lib.mxdet_roi_align_bwd_gather_prepare(C.byref(d), dmaps[0].shape[0], ptr(rois), ptr(levels), rois.shape[0],
                                                 pooled[0], pooled[1], sampling_ratio, ptr(workspace), workspace.numel(),
                                                 stream_ptr()), "roi_align_bwd_gather_prepare")


def roi_align_backward_gather(dmaps, scales, rois, levels, grad_out, sampling_ratio=2, lvl_min=2, accumulate=False,
                              workspace=None, prepared=False):
    """Deterministic gather form: writes (or adds to) the bf16 gradient maps dmaps[l] ([N,H,W,C]) directly.
    prepared: the records are already in `workspace` (roi_align_backward_gather_prepare)."""
    lib = _lib.load()
    N, C_ = dmaps[0].shape[0], dmaps[0].shape[3]
    R, PH, PW = grad_out.shape[0], grad_out.shape[1], grad_out.shape[2]
    d = _pyr(dmaps, scales, lvl_min)
    assert workspace is not None or not prepared
    workspace = _gather_workspace(lib, d, N, R, dmaps[0].device, workspace)
    fn = lib.mxdet_roi_align_bwd_gather_prepared if prepared else lib.mxdet_roi_align_bwd_gather
    check(fn(C.byref(d), N, C_, ptr(rois), ptr(levels), R, PH, PW, sampling_ratio,
             ptr(grad_out), int(accumulate), ptr(workspace), workspace.numel(), stream_ptr()),
          "roi_align_bwd_gather")
