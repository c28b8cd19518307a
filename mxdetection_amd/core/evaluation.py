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
