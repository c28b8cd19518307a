"""ops: pyramid proposal generation (MXNet role: contrib.Proposal / MultiProposal / pyramid-proposal CustomOp)."""
import ctypes as C

import torch

from .. import _lib
from .._lib import PyramidT, check, ptr, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


class PyramidProposal:
    """Holds the level geometry, device base anchors and the caller-owned workspace."""

    def __init__(self, base_anchors, strides, pre_nms_top_n=2000, post_nms_top_n=2000, nms_thresh=0.7, min_size=0.0):
        self.base = [b.contiguous() for b in base_anchors]  # per level [A,4] cuda f32
        self.strides = list(strides)
        self.A = self.base[0].shape[0]
        self.pre, self.post, self.thresh, self.min_size = pre_nms_top_n, post_nms_top_n, nms_thresh, min_size
        self._ws = None
        self._retired = []

    def _desc(self, cls, reg, layout):
        """cls[l]/reg[l]: 'nhwc_fused' -> one tensor [N,H,W,Cpad] (logits ch 0..A-1, deltas A..5A-1);
        'nchw' -> cls [N,A,H,W], reg [N,4A,H,W]."""
        d = PyramidT()
        L = len(cls)
        d.num_levels, d.A = L, self.A
        d.dtype = _DT[cls[0].dtype]
        for l in range(L):
            c, r = cls[l], reg[l]
            if layout == "nhwc_fused":
                N, H, W, Cp = c.shape
                d.cls_sn[l], d.cls_sy[l], d.cls_sx[l], d.cls_sa[l] = H * W * Cp, W * Cp, Cp, 1
                d.reg_sn[l], d.reg_sy[l], d.reg_sx[l], d.reg_sc[l] = H * W * Cp, W * Cp, Cp, 1
                d.cls[l] = c.data_ptr()
                d.reg[l] = c.data_ptr() + self.A * c.element_size()
            else:
                N, _, H, W = c.shape
                d.cls_sn[l], d.cls_sy[l], d.cls_sx[l], d.cls_sa[l] = self.A * H * W, W, 1, H * W
                d.reg_sn[l], d.reg_sy[l], d.reg_sx[l], d.reg_sc[l] = 4 * self.A * H * W, W, 1, H * W
                d.cls[l] = c.data_ptr()
                d.reg[l] = r.data_ptr()
            d.H[l], d.W[l], d.stride[l] = H, W, self.strides[l]
            d.base_anchors[l] = self.base[l].data_ptr()
        return d, N

    def __call__(self, cls, reg, im_info, layout="nhwc_fused", out=None):
        lib = _lib.load()
        d, N = self._desc(cls, reg, layout)
        dev = im_info.device
        need = lib.mxdet_proposal_workspace_bytes(C.byref(d), N, self.pre)
        if self._ws is None or self._ws.numel() < need:
            self._retired.append(self._ws)     # a captured step may hold the old one by address: never freed
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        if out is None:
            rois = torch.empty((N, self.post, 5), dtype=torch.float32, device=dev)
            scores = torch.empty((N, self.post), dtype=torch.float32, device=dev)
            anchor = torch.empty((N, self.post), dtype=torch.int32, device=dev)
            num = torch.empty((N,), dtype=torch.int32, device=dev)
        else:
            rois, scores, anchor, num = out
        check(lib.mxdet_proposal(C.byref(d), N, ptr(im_info), self.pre, self.post, self.thresh, self.min_size,
                                 ptr(rois), ptr(scores), ptr(anchor), ptr(num), ptr(self._ws), self._ws.numel(),
                                 stream_ptr()), "proposal")
        return rois, scores, anchor, num
