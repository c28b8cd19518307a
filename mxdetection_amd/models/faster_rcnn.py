"""Faster R-CNN ResNet-FPN: one training step = forward, targets, losses, explicit backward, gradient
all-reduce (RCCL, overlapped with backward), SGD-momentum update. Every arithmetic step is a hand-written
HIP kernel reached through the C-ABI; torch supplies device memory, streams and torch.distributed only.

Assembles the plugin parts the reference declares (/root/reference/README.md:27-32). The training loop
itself is not declared anywhere in the reference tree; this is the build's own.
"""
import os

import torch

from .backbones import ResNet
from .bbox_heads import BBoxHead
from .mask_heads import FCNMaskHead
from .necks import FPN
from .roi_extractors import FPNRoIExtractor
from .rpn_heads import RPNHead
from .utils.detector import DetectorBase
from .utils.layers import Workspace


class FasterRCNN(DetectorBase):
    def __init__(self, device="cuda", depth=50, num_classes=81, seed=7, rpn_seed=99, rois_per_image=512,
                 pre_nms_top_n=2000, post_nms_top_n=2000, with_mask=False):
        gen = torch.Generator().manual_seed(seed)
        self._init_base(device)
        self.strides = [4, 8, 16, 32, 64]
        # registration order == backward completion order (buckets become final early)
        self.with_mask = with_mask
        # RoIAlign backward: deterministic gather form (no atomics, no fp32 accumulators, writes the bf16 maps directly);
        # False selects the fp32-atomic scatter form (310 us + zero-fill + finalize at the benchmark shape)
        self.roi_bwd_gather = True
        # anchor assignment at the start of the step, on the branch stream (see forward_backward): +0.9 % on the step
        self.early_anchor_targets = os.environ.get("MXDET_TUNE_EARLY_ANCHORS", "1") == "1"
        self.mask_head = None
        if with_mask:   # Mask R-CNN (BASELINE.json config 4): the mask branch's backward runs first
            self.mask_head = FCNMaskHead(256, self.arena, self.ws, device, gen, num_classes=num_classes,
                                         rois_per_image=max(1, rois_per_image // 4))
            self.mask_roi_extractor = FPNRoIExtractor([4, 8, 16, 32], pooled=(14, 14), device=device)
        self.mark_mask = self.arena.size
        self.bbox_head = BBoxHead(7 * 7 * 256, self.arena, self.ws, device, gen, num_classes=num_classes,
                                  rois_per_image=rois_per_image, seed=rpn_seed)
        self.mark_head = self.arena.size
        # own wgrad scratch: the RPN training branch runs on its own stream (enable_branch_stream) and must not
        # share the split-K slabs with the weight-gradient stream of the rest of the model
        self.ws_rpn = Workspace(device)
        self.rpn_head = RPNHead(256, self.strides, self.arena, self.ws_rpn, device, gen, pre_nms_top_n=pre_nms_top_n,
                                post_nms_top_n=post_nms_top_n, seed=rpn_seed)
        self.mark_rpn = self.arena.size
        self.neck = FPN([256, 512, 1024, 2048], 256, self.arena, self.ws, device, gen)
        self.mark_fpn = self.arena.size
        self.backbone = ResNet(depth, self.arena, self.ws, device, gen)
        self.roi_extractor = FPNRoIExtractor(self.strides[:4], device=device)
        layers = self.bbox_head.layers() + self.rpn_head.layers() + self.neck.layers() + self.backbone.layers()
        if with_mask:
            layers = self.mask_head.layers() + layers
        self._finalize_params(layers)
        # arena offsets after each backbone stage (layer4, layer3, layer2) for bucketed all-reduce
        self.stage_marks = {}
        for si in (3, 2, 1):
            last = self.backbone.stages[si][0].layers()[-1]
            e = self.arena.entries[last.wi]
            self.stage_marks[si] = e[2] + (e[3] + 63) // 64 * 64

    def plan(self, N, H, W, g_max):
        key = (N, H, W, g_max)
        if self.planned == key:
            return
        self._guard_replan(key)
        dev = self.device
        c_shapes = self.backbone.plan((N, 3, H, W))
        self.neck.plan(c_shapes)
        p_shapes = [(s[0], s[1], s[2], 256) for s in c_shapes]
        p_shapes.append((N, (p_shapes[-1][1] + 1) // 2, (p_shapes[-1][2] + 1) // 2, 256))
        self.rpn_head.plan(p_shapes, g_max)
        self.bbox_head.plan(N)
        if self.with_mask:
            self.mask_head.plan(N)
        self.ws.get()
        self.ws_rpn.get()
        # one flat bf16 buffer (levels are views, finest first): the RoI extractor finalizes P2..P5 with one launch
        sizes = [s[0] * s[1] * s[2] * s[3] for s in p_shapes]
        self.dP_flat = torch.empty((sum(sizes),), dtype=torch.bfloat16, device=dev)
        self.dP, off = [], 0
        for s, n in zip(p_shapes, sizes):
            self.dP.append(self.dP_flat[off:off + n].view(s))
            off += n
        self.dC = [None] + [torch.empty(s, dtype=torch.bfloat16, device=dev) for s in c_shapes[1:]]
        self.planned = key

    def forward_backward(self, image, gt_boxes, im_info, step=0, image_offset=0, step_dev=None, gt_masks=None):
        """image NCHW [N,3,H,W]; gt_boxes [N,G,5] f32 (class < 0 padding); im_info [N,3] f32."""
        N, _, H, W = image.shape
        self.plan(N, H, W, gt_boxes.shape[1])
        early = self.early_anchor_targets and self.branch is not None
        if early:
            # anchor assignment needs the ground truth only: on the branch stream it runs underneath the backbone forward,
            # and the RPN branch proper starts with its losses and heavy convolutions instead of ~100 us of small kernels
            with self._branch_ctx():
                self.rpn_head.assign_targets(gt_boxes, im_info, step, image_offset, step_dev)
        C = self.backbone.forward(image)
        P = self.neck.forward(C)
        self.rpn_head.forward(P)
        # The RPN training branch (anchor targets, RPN losses, RPN head backward: MFMA-heavy) depends only on the
        # head outputs; the proposal -> RoI -> box-head chain (long, low-occupancy selection/NMS kernels) does not
        # depend on it. They run on two streams and meet at dP.
        # The RPN head's weight gradients leave the branch and run on the side stream, with the head bucket, underneath
        # the RoIAlign gather (+0.8 % on the step).
        defer_rpn = (self.roi_bwd_gather and self.ws.side is not None and self.ws_rpn.grouping and self.ws.grouping
                     and self._bucket_here(0) and os.environ.get("MXDET_TUNE_DEFER_RPN_WGRAD", "1") == "1")
        with self._branch_ctx():
            rpn_loss = self.rpn_head.loss_and_grad(gt_boxes, im_info, step, image_offset, step_dev=step_dev, assigned=early)
            self.rpn_head.backward(self.dP, [False] * 5, flush=not defer_rpn)
        rois, _, _, num_rois = self.rpn_head.get_proposals(im_info)
        rois_s = self.bbox_head.sample(rois, num_rois, gt_boxes, step, image_offset, step_dev)
        pooled = self.roi_extractor.forward(P, rois_s, prepare_gather=self.roi_bwd_gather)
        self.bbox_head.forward(pooled)
        rcnn_loss = self.bbox_head.loss_and_grad()
        mask_loss = None
        if self.with_mask:
            mrois = self.mask_head.select_rois(self.bbox_head)
            self.mask_head.targets(gt_masks)
            mpooled = self.mask_roi_extractor.forward(P, mrois, prepare_gather=self.roi_bwd_gather)
            self.mask_head.forward(mpooled)
            mask_loss = self.mask_head.loss_and_grad()
        # ---- backward ----
        if self.with_mask:
            d_mpooled = self.mask_head.backward()
        d_pooled = self.bbox_head.backward()
        if self.roi_bwd_gather:
            self._join_branch()                                       # dP[l] holds the RPN part
            # (defer_rpn: the RPN head's weight gradients -- MFMA-bound -- go to the side stream with this bucket instead of
            # lengthening the branch.) The gather is issued FIRST and the head bucket's weight gradients, update and filter
            # transposes behind it (round 3: with the three-tap weight-gradient kernel the gather beside a 2,000-workgroup
            # MFMA grid took 235-310 us instead of its 70-90 us alone; alone first, then the bucket beside the P2 data
            # gradients, is +0.3 % on the step. MXDET_TUNE_ROI_FIRST=0: the round-2 order, bucket first)
            lo = 0
            roi_first = os.environ.get("MXDET_TUNE_ROI_FIRST", "1") == "1"     # the gather first, alone; the bucket behind it
            if self._bucket_here(0) and not roi_first:
                self._reduce(0, self.mark_rpn, pre=self.ws_rpn if defer_rpn else None)
                lo = self.mark_rpn
            self.roi_extractor.backward_gather(d_pooled.view(pooled.shape), self.dP[:4], accumulate=True)
            if self.with_mask:
                self.mask_roi_extractor.backward_gather(d_mpooled, self.dP[:4], accumulate=True)
            if self._bucket_here(0) and roi_first:
                self._reduce(0, self.mark_rpn, pre=self.ws_rpn if defer_rpn else None)
                lo = self.mark_rpn
        else:
            acc = self.roi_extractor.backward(d_pooled.view(pooled.shape), self.dP[:4], finalize=False)
            if self.with_mask:
                self.mask_roi_extractor.backward(d_mpooled, self.dP[:4], shared_acc=acc, zero=False, finalize=False)
            self._join_branch()
            self.roi_extractor.finalize(self.dP[:4], accumulate=True)     # dP[l] = RPN part + RoI part
            self._reduce(0, self.mark_rpn)
            lo = self.mark_rpn
        self.neck.backward(self.dP, self.dC, [False, True, True, True])
        if self._bucket_here(1):
            self._reduce(lo, self.mark_fpn)
            lo = self.mark_fpn
        # where the next step's frozen front end may start: behind the backward of backbone stage `tail_at` (1 = where the
        # data-gradient chain ends; 2 / 3 = one / two stages earlier, 4 = before the backbone's backward)
        tail_at = int(os.environ.get("MXDET_TUNE_TAIL_AT", "1"))
        if tail_at >= 4:
            self._mark_tail()
        for si in (3, 2, 1):
            self._backbone_stage_backward(si)
            if si == tail_at:
                self._mark_tail()
            if si == 1 or self._bucket_here(5 - si):       # reduce points 2 (layer4), 3 (layer3); layer2 always closes
                self._reduce(lo, self.stage_marks[si])
                lo = self.stage_marks[si]
        if self.with_mask:
            return rpn_loss, rcnn_loss, mask_loss
        return rpn_loss, rcnn_loss

    def predict(self, image, im_info, score_thresh=0.05, nms_thresh=0.5, max_per_image=100, with_masks=False,
                mask_thresh=0.5):
        """Inference: forward, proposals, box head, then softmax / decode / per-class NMS / top-k on the GPU
        (core/evaluation). Returns (dets [N,max_per_image,6] = x1,y1,x2,y2,score,class;
        num_dets [N]); with_masks (Mask R-CNN) adds the pasted-back instance masks [N,max_per_image,H,W] u8 in the
        frame of the network input: mask head on the detected boxes, sigmoid, bilinear resize into the box, threshold."""
        from ..core.evaluation import DetectionPostprocess
        N, _, H, W = image.shape
        g_max = self.planned[3] if self.planned is not None and self.planned[:3] == (N, H, W) else 100
        self.plan(N, H, W, g_max)
        P = self.neck.forward(self.backbone.forward(image))
        self.rpn_head.forward(P)
        rois, _, _, num_rois = self.rpn_head.get_proposals(im_info)
        pooled = self.roi_extractor.forward(P, rois.view(-1, 5))
        o = self.bbox_head.forward(pooled)
        o2 = o.view(o.shape[0], -1)
        key = (score_thresh, nms_thresh, max_per_image)
        if getattr(self, "_post_key", None) != key:
            self._post = DetectionPostprocess(self.bbox_head.nc, score_thresh, nms_thresh, max_per_image,
                                              stds=self.bbox_head.stds)
            self._post_key = key
        dets, num = self._post(o2[:, :self.bbox_head.nc], o2[:, self.bbox_head.nc:], rois.view(-1, 5), num_rois, im_info)
        if not with_masks:
            return dets, num
        assert self.with_mask, "with_masks needs a model built with the mask head"
        from ..core import mask as M_
        M = dets.shape[1]
        flat = dets.view(N * M, 6)
        drois = torch.empty((N * M, 5), dtype=torch.float32, device=dets.device)
        drois[:, 0] = torch.arange(N, device=dets.device, dtype=torch.float32).repeat_interleave(M)
        drois[:, 1:] = flat[:, :4]            # padding rows are zero boxes with class -1: pasted as empty masks
        mpooled = self.mask_roi_extractor.forward(P, drois)
        logits = self.mask_head.forward(mpooled)
        masks = M_.mask_paste(logits, flat, H, W, mask_thresh)
        return dets, num, masks.view(N, M, H, W)

    def _backbone_stage_backward(self, si):
        stage = self.backbone.stages[si]
        ds = self.dC[si]
        for bi in reversed(range(len(stage))):
            b = stage[bi]
            if bi > 0:
                ds = b.backward(ds, b._buf("dx", b.x.shape), False)
            elif b.need_dx:
                b.backward(ds, self.dC[si - 1], True)
            else:
                b.backward(ds, None, False)
