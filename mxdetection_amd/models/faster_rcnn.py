"""Faster R-CNN ResNet-FPN: one training step = forward, targets, losses, explicit backward, gradient
all-reduce (RCCL, overlapped with backward), SGD-momentum update. Every arithmetic step is a hand-written
HIP kernel reached through the C-ABI; torch supplies device memory, streams and torch.distributed only.

Assembles the plugin parts the reference declares (/root/reference/README.md:27-32). The training loop
itself is not declared anywhere in the reference tree; this is the build's own.
"""
import torch

from .backbones import ResNet
from .bbox_heads import BBoxHead
from .mask_heads import FCNMaskHead
from .necks import FPN
from .roi_extractors import FPNRoIExtractor
from .rpn_heads import RPNHead
from .utils.dp import BucketReducer
from .utils.layers import ParamArena, Workspace


class FasterRCNN:
    def __init__(self, device="cuda", depth=50, num_classes=81, seed=7, rpn_seed=99, rois_per_image=512,
                 pre_nms_top_n=2000, post_nms_top_n=2000, with_mask=False):
        gen = torch.Generator().manual_seed(seed)
        self.device = device
        self.arena = ParamArena(device)
        self.ws = Workspace(device)
        self.strides = [4, 8, 16, 32, 64]
        # registration order == backward completion order (buckets become final early)
        self.with_mask = with_mask
        self.mask_head = None
        if with_mask:   # Mask R-CNN (BASELINE.json config 4): the mask branch's backward runs first
            self.mask_head = FCNMaskHead(256, self.arena, self.ws, device, gen, num_classes=num_classes,
                                         rois_per_image=max(1, rois_per_image // 4))
            self.mask_roi_extractor = FPNRoIExtractor([4, 8, 16, 32], pooled=(14, 14), device=device)
        self.mark_mask = self.arena.size
        self.bbox_head = BBoxHead(7 * 7 * 256, self.arena, self.ws, device, gen, num_classes=num_classes,
                                  rois_per_image=rois_per_image, seed=rpn_seed)
        self.mark_head = self.arena.size
        self.rpn_head = RPNHead(256, self.strides, self.arena, self.ws, device, gen, pre_nms_top_n=pre_nms_top_n,
                                post_nms_top_n=post_nms_top_n, seed=rpn_seed)
        self.mark_rpn = self.arena.size
        self.neck = FPN([256, 512, 1024, 2048], 256, self.arena, self.ws, device, gen)
        self.mark_fpn = self.arena.size
        self.backbone = ResNet(depth, self.arena, self.ws, device, gen)
        self.roi_extractor = FPNRoIExtractor(self.strides[:4], device=device)
        self.arena.finalize()
        self.layers = self.bbox_head.layers() + self.rpn_head.layers() + self.neck.layers() + self.backbone.layers()
        if with_mask:
            self.layers = self.mask_head.layers() + self.layers
        for l in self.layers:
            l.materialize()
        self.arena.refresh_bf16()
        self.refresh_transposed()
        # arena offsets after each backbone stage (layer4, layer3, layer2) for bucketed all-reduce
        self.stage_marks = {}
        for si in (3, 2, 1):
            last = self.backbone.stages[si][0].layers()[-1]
            e = self.arena.entries[last.wi]
            self.stage_marks[si] = e[2] + (e[3] + 63) // 64 * 64
        self.planned = None
        self.dist = None
        self.world = 1
        self.reducer = BucketReducer(self.arena.g, None)
        self.segments = None
        self._cap = False
        self._cur_graph = None

    def export_params(self):
        """name -> fp32 CPU tensor of every parameter as the kernels see it (bf16 filters, fp32 biases)."""
        out = {"stem.weight": self.backbone.stem_w.float().cpu(), "stem.bias": self.backbone.stem_b.float().cpu()}
        frozen = [l for st in self.backbone.stages for b in st for l in b.layers() if not l.trainable]
        for l in self.layers + frozen:
            out[l.name + ".weight"] = l.w_bf16.float().cpu()
            if l.has_bias:
                out[l.name + ".bias"] = l.bias_f32.float().cpu()
        return out

    def export_grads(self):
        """name -> fp32 CPU gradient of every trainable parameter."""
        return {e[0]: self.arena.view(i, "g").float().cpu() for i, e in enumerate(self.arena.entries)}

    def num_params(self):
        return sum(e[3] for e in self.arena.entries)

    def refresh_transposed(self):
        """[Cout,KH,KW,Cin] -> [Cin,KH,KW,Cout] copies for dgrad: one batched launch for all trainable filters."""
        from ..ops import dense
        if getattr(self, "_tr_table", None) is None:
            pairs = [(l.w_bf16, l.wt) for l in self.layers if l.trainable]
            self._tr_table = dense.make_transpose_table(pairs, self.device)
        dense.filter_transpose_batched(*self._tr_table)

    def enable_wgrad_stream(self):
        """Issue weight-gradient kernels on a second stream (overlaps them with the data-gradient chain)."""
        self.ws.side = torch.cuda.Stream()

    def enable_data_parallel(self, world_size):
        import torch.distributed as dist
        self.dist = dist
        self.world = world_size
        self.reducer = BucketReducer(self.arena.g, dist)

    def plan(self, N, H, W, g_max):
        key = (N, H, W, g_max)
        if self.planned == key:
            return
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
        self.dP = [torch.empty(s, dtype=torch.bfloat16, device=dev) for s in p_shapes]
        self.dC = [None] + [torch.empty(s, dtype=torch.bfloat16, device=dev) for s in c_shapes[1:]]
        self.planned = key

    # ---- gradient buckets -----------------------------------------------------------------------
    def _reduce(self, lo, hi):
        self.ws.join()            # the bucket's weight gradients were produced on the side stream
        if self._cap:
            if self.dist is not None and hi > lo:   # cut the graph here: the all-reduce runs between segments
                self._seg_end()
                self.segments.append(("reduce", lo, hi))
                self._seg_begin()
            return
        self.reducer.reduce(lo, hi)

    # ---- hipGraph capture of the whole step (static shapes): removes ~450 host launches per step ----
    def _seg_begin(self):
        self._cur_graph = torch.cuda.CUDAGraph()
        self._cur_graph.capture_begin(pool=self._pool)

    def _seg_end(self):
        self._cur_graph.capture_end()
        self.segments.append(self._cur_graph)
        self._cur_graph = None

    def capture(self, image, gt_boxes, im_info, lr, image_offset=0, warmup=2, gt_masks=None):
        """Capture forward+backward+update into hipGraph segments (cut only at gradient all-reduces).
        The RNG step counter is read from device memory (step_dev), inputs from static buffers."""
        dev = self.device
        self.static_in = (image.clone(), gt_boxes.clone(), im_info.clone())
        self.static_masks = gt_masks.clone() if gt_masks is not None else None
        self.step_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        for i in range(warmup):     # eager warm-up: plans shapes and allocates every buffer
            self.train_step(*self.static_in, step=i, image_offset=image_offset, lr=lr, gt_masks=self.static_masks)
        torch.cuda.synchronize()
        self._pool = torch.cuda.graph_pool_handle()
        self.segments = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._cap = True
            self._seg_begin()
            losses = self.forward_backward(*self.static_in, step=0, image_offset=image_offset, step_dev=self.step_dev,
                                           gt_masks=self.static_masks)
            self.optimizer_step(lr)
            self._seg_end()
            self._cap = False
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.static_losses = losses

    def replay(self, image, gt_boxes, im_info, step, gt_masks=None):
        """One training step from the captured graphs."""
        si = self.static_in
        if image is not si[0]:
            si[0].copy_(image, non_blocking=True)
            si[1].copy_(gt_boxes, non_blocking=True)
            si[2].copy_(im_info, non_blocking=True)
            if gt_masks is not None:
                self.static_masks.copy_(gt_masks, non_blocking=True)
        self.step_dev.fill_(step)
        for seg in self.segments:
            if isinstance(seg, tuple):
                if seg[0] == "reduce":
                    self.reducer.reduce(seg[1], seg[2])
                else:
                    self.reducer.wait()
            else:
                seg.replay()
        return self.static_losses

    def forward_backward(self, image, gt_boxes, im_info, step=0, image_offset=0, step_dev=None, gt_masks=None):
        """image NCHW [N,3,H,W]; gt_boxes [N,G,5] f32 (class < 0 padding); im_info [N,3] f32."""
        N, _, H, W = image.shape
        self.plan(N, H, W, gt_boxes.shape[1])
        C = self.backbone.forward(image)
        P = self.neck.forward(C)
        self.rpn_head.forward(P)
        rois, _, _, num_rois = self.rpn_head.get_proposals(im_info)
        rpn_loss = self.rpn_head.loss_and_grad(gt_boxes, im_info, step, image_offset, step_dev=step_dev)
        rois_s = self.bbox_head.sample(rois, num_rois, gt_boxes, step, image_offset, step_dev)
        pooled = self.roi_extractor.forward(P, rois_s)
        self.bbox_head.forward(pooled)
        rcnn_loss = self.bbox_head.loss_and_grad()
        mask_loss = None
        if self.with_mask:
            mrois = self.mask_head.select_rois(self.bbox_head)
            self.mask_head.targets(gt_masks)
            mpooled = self.mask_roi_extractor.forward(P, mrois)
            self.mask_head.forward(mpooled)
            mask_loss = self.mask_head.loss_and_grad()
        # ---- backward ----
        if self.with_mask:
            d_mpooled = self.mask_head.backward()
            self._reduce(0, self.mark_mask)
        d_pooled = self.bbox_head.backward()
        self._reduce(self.mark_mask, self.mark_head)
        if self.with_mask:
            acc = self.roi_extractor.backward(d_pooled.view(pooled.shape), self.dP[:4], finalize=False)
            self.mask_roi_extractor.backward(d_mpooled, self.dP[:4], shared_acc=acc, zero=False, finalize=True)
        else:
            self.roi_extractor.backward(d_pooled.view(pooled.shape), self.dP[:4])
        self.rpn_head.backward(self.dP, [True, True, True, True, False])
        self._reduce(self.mark_head, self.mark_rpn)
        self.neck.backward(self.dP, self.dC, [False, True, True, True])
        self._reduce(self.mark_rpn, self.mark_fpn)
        lo = self.mark_fpn
        for si in (3, 2, 1):
            self._backbone_stage_backward(si)
            self._reduce(lo, self.stage_marks[si])
            lo = self.stage_marks[si]
        if self.with_mask:
            return rpn_loss, rcnn_loss, mask_loss
        return rpn_loss, rcnn_loss

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

    def optimizer_step(self, lr, momentum=0.9, wd=1e-4):
        self.ws.join()
        if self._cap:
            if self.dist is not None:
                self._seg_end()
                self.segments.append(("wait",))
                self._seg_begin()
        else:
            self.reducer.wait()
        rescale = 1.0 / self.world
        self.arena.sgd_step(lr, momentum, wd, rescale)
        self.refresh_transposed()

    def train_step(self, image, gt_boxes, im_info, step=0, image_offset=0, lr=0.0025, gt_masks=None):
        losses = self.forward_backward(image, gt_boxes, im_info, step, image_offset, gt_masks=gt_masks)
        self.optimizer_step(lr)
        return losses
