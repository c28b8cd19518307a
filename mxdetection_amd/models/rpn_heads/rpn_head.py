"""Region Proposal Network head over an FPN pyramid: shared 3x3 conv + ReLU, one fused 1x1 conv producing the
objectness logits (channels 0..A-1) and box deltas (channels A..5A-1), RPN targets + losses, proposal generation.

Plugin slot: models/rpn_heads (/root/reference/README.md:28) with ops (README.md:24) and core/anchor, core/loss
(README.md:16,19). Everything (targets, sampling, NMS) stays on the GPU; MXNet-lineage code runs these as
numpy/Cython CustomOps on the host.
"""
import os

import torch

from ...core import anchor as A_
from ...core import loss as L_
from ...ops import dense
from ...ops.proposal import PyramidProposal
from ..utils.layers import ConvLayer, cached_buf

HEAD_CPAD = 64   # fused cls+reg output channels padded so that dgrad's reduction dim is a multiple of 64


RELU_BITS = os.environ.get("MXDET_TUNE_RELU_BITS", "1") == "1"


class RPNHead:
    def __init__(self, channels, strides, arena, ws, device, gen, ratios=(0.5, 1.0, 2.0), scales=(8,),
                 pre_nms_top_n=2000, post_nms_top_n=2000, nms_thresh=0.7, min_size=0.0, batch_size=256,
                 fg_fraction=0.5, fg_thresh=0.7, bg_thresh=0.3, sigma=3.0, seed=99):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        self.A = len(ratios) * len(scales)
        assert 5 * self.A <= HEAD_CPAD
        self.out = ConvLayer("rpn.out", channels, HEAD_CPAD, 1, init_std=0.01, cout_real=5 * self.A, **kw)
        self.conv = ConvLayer("rpn.conv", channels, channels, 3, init_std=0.01, **kw)
        self.strides = list(strides)
        self.base = [torch.from_numpy(A_.generate_base_anchors(s, ratios, scales)).to(device) for s in strides]
        self.proposal = PyramidProposal(self.base, strides, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size)
        self.batch_size, self.fg_fraction, self.fg_thresh, self.bg_thresh = batch_size, fg_fraction, fg_thresh, bg_thresh
        self.sigma, self.seed, self.device, self.C = sigma, seed, device, channels
        self.bufs = {}
        self.anchors = None
        self.level_shapes = None

    def layers(self):
        return [self.out, self.conv]

    def _buf(self, key, shape, dtype=torch.bfloat16, zero=False):
        return cached_buf(self.bufs, key, shape, dtype, self.device, zero)

    def plan(self, p_shapes, g_max):
        for s in p_shapes:
            self.conv.plan(s)
            self.out.plan(s)
        self.level_shapes = [(s[1], s[2]) for s in p_shapes]
        N = p_shapes[0][0]
        self.anchors = torch.cat([A_.generate_anchors(self.base[l], H, W, self.strides[l])
                                  for l, (H, W) in enumerate(self.level_shapes)])
        self.level_offsets = [0]
        for (H, W) in self.level_shapes:
            self.level_offsets.append(self.level_offsets[-1] + H * W * self.A)
        At = self.anchors.shape[0]
        self.at_ws = A_.AnchorTargetWorkspace(N, At, g_max, self.device)
        self.at_out = (torch.empty((N, At), dtype=torch.int32, device=self.device),
                       torch.empty((N, At), dtype=torch.int32, device=self.device),
                       torch.empty((N, At, 4), dtype=torch.float32, device=self.device),
                       torch.empty((N, At), dtype=torch.float32, device=self.device))
        self.nparts = [L_.rpn_loss_num_partials(N, H, W) for (H, W) in self.level_shapes]
        self.partial = torch.zeros((2 * sum(self.nparts),), dtype=torch.float32, device=self.device)
        self.loss = torch.zeros((2,), dtype=torch.float32, device=self.device)

    def forward(self, P):
        """The largest level keeps its own launches (it fills the chip and has its own tile path); the remaining levels
        -- a few dozen to a few hundred workgroups each -- share one grouped launch per layer."""
        self.P = P
        tb = [self._buf("t%d" % l, p.shape) for l, p in enumerate(P)]
        hb = [self._buf("h%d" % l, p.shape[:3] + (HEAD_CPAD,)) for l, p in enumerate(P)]
        # 1-bit ReLU masks of the conv activations for the head's data gradient (the P2-level activation is 68.8 MB)
        self.tbits = [self._buf("tb%d" % l, p.shape[:3] + (p.shape[3] // 8,), dtype=torch.uint8) if RELU_BITS else None
                      for l, p in enumerate(P)]
        self.conv.forward(P[0], relu=True, out=tb[0], bits_out=self.tbits[0])
        dense.conv2d_group("fwd", [self.conv.fwd_call(P[l], relu=True, out=tb[l], bits_out=self.tbits[l])
                                   for l in range(1, len(P))], self.device)
        self.out.forward(tb[0], out=hb[0])
        dense.conv2d_group("fwd", [self.out.fwd_call(tb[l], out=hb[l]) for l in range(1, len(P))], self.device)
        self.t, self.h = tb, hb
        return self.h

    def get_proposals(self, im_info):
        return self.proposal(self.h, self.h, im_info, layout="nhwc_fused")

    def assign_targets(self, gt_boxes, im_info, step, image_offset, step_dev=None):
        """Anchor labels / regression targets. They depend on the ground truth and the anchor grid only -- not on any
        network output -- so the caller may issue this long before the head has run."""
        labels, _, targets, _ = A_.assign_anchor(self.anchors, gt_boxes, im_info, self.fg_thresh, self.bg_thresh, 0.0,
                                                 self.batch_size, self.fg_fraction, self.seed, step, image_offset,
                                                 self.at_ws, self.at_out, step_dev)
        self._assigned = (labels, targets)

    def loss_and_grad(self, gt_boxes, im_info, step, image_offset, loss_scale=1.0, step_dev=None, assigned=False):
        """Assign anchors (unless assign_targets() already ran for this step), compute the RPN losses and
        d(loss)/d(head) for every level (bf16)."""
        if not assigned:
            self.assign_targets(gt_boxes, im_info, step, image_offset, step_dev)
        labels, targets = self._assigned
        N = gt_boxes.shape[0]
        norm = 1.0 / float(N * self.batch_size)
        self.gh = []
        off = 0
        for l, h in enumerate(self.h):
            g = self._buf("gh%d" % l, h.shape)
            L_.rpn_loss_level(h, self.A, labels, targets, self.level_offsets[l], self.sigma, norm, loss_scale, g,
                              self.partial[2 * off:])
            off += self.nparts[l]
            self.gh.append(g)
        L_.loss_finalize(self.partial, off, 2, self.loss)
        return self.loss

    def backward(self, dP, dP_has_grad, flush=True):
        """Adds the RPN branch's gradient into dP[l] (overwrites where dP_has_grad[l] is False). flush=False leaves the
        recorded weight gradients pending in the workspace: the caller issues them later (ws.flush())."""
        L = len(self.h)
        dt = [self._buf("dt%d" % l, self.t[l].shape) for l in range(L)]
        grouped = self.out.ws.grouping     # grouped form: the plan sums the levels of a shared filter itself
        for l in range(L):
            self.out.backward_weight(self.t[l], self.gh[l], accumulate=(l > 0) and not grouped)
        self.out.backward_data(self.gh[0], self.t[0].shape, relu_mask=self.t[0], out=dt[0], relu_bits=self.tbits[0])
        dense.conv2d_group("dgrad", [self.out.dgrad_call(self.gh[l], self.t[l].shape, relu_mask=self.t[l], out=dt[l],
                                                         relu_bits=self.tbits[l])
                                     for l in range(1, L)], self.device)
        for l in range(L):
            self.conv.backward_weight(self.P[l], dt[l], accumulate=(l > 0) and not grouped)
        self.conv.backward_data(dt[0], self.P[0].shape, accumulate=dP_has_grad[0], out=dP[0])
        dense.conv2d_group("dgrad", [self.conv.dgrad_call(dt[l], self.P[l].shape, accumulate=dP_has_grad[l], out=dP[l])
                                     for l in range(1, L)], self.device)
        if flush:
            self.out.ws.flush()
