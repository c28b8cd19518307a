"""RetinaNet ResNet-FPN (BASELINE.json config 5: ResNet-101-FPN, dense 9-anchor heads, focal loss): one training step
on hand-written HIP kernels, same arena / graph / data-parallel machinery as Faster R-CNN."""
import torch

from .backbones import ResNet
from .necks.fpn import RetinaFPN
from .rpn_heads.retina_head import RetinaHead
from .utils.detector import DetectorBase


class RetinaNet(DetectorBase):
    def __init__(self, device="cuda", depth=101, num_classes=80, seed=7):
        gen = torch.Generator().manual_seed(seed)
        self._init_base(device)
        self.strides = [8, 16, 32, 64, 128]
        self.head = RetinaHead(256, self.strides, self.arena, self.ws, device, gen, num_classes=num_classes)
        self.mark_head = self.arena.size
        self.neck = RetinaFPN([512, 1024, 2048], 256, self.arena, self.ws, device, gen)
        self.mark_fpn = self.arena.size
        self.backbone = ResNet(depth, self.arena, self.ws, device, gen)
        self._finalize_params(self.head.layers() + self.neck.layers() + self.backbone.layers())
        self.head.post_materialize()
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
        c_shapes = self.backbone.plan((N, 3, H, W))
        p_shapes = self.neck.plan(c_shapes[1:])
        self.head.plan(p_shapes, g_max)
        self.ws.get()
        dev = self.device
        self.dP = [torch.empty(s, dtype=torch.bfloat16, device=dev) for s in p_shapes]
        self.dC = [None] + [torch.empty(s, dtype=torch.bfloat16, device=dev) for s in c_shapes[1:]]
        self.planned = key

    def predict(self, image, im_info, score_thresh=0.05, nms_thresh=0.5, max_per_image=100, pre_nms_top_n=1000):
        """Inference: forward, then per-level top-k / decode / per-class NMS / top-k on the GPU (core/evaluation
        RetinaDetect). Returns (dets [N,max_per_image,6] = x1,y1,x2,y2,score,class in 1..C; num_dets [N])."""
        from ..core.evaluation import RetinaDetect
        N, _, H, W = image.shape
        g_max = self.planned[3] if self.planned is not None and self.planned[:3] == (N, H, W) else 100
        self.plan(N, H, W, g_max)
        P = self.neck.forward(self.backbone.forward(image)[1:])
        co, bo = self.head.forward(P)
        key = (score_thresh, nms_thresh, max_per_image, pre_nms_top_n)
        if getattr(self, "_det_key", None) != key:
            self._det = RetinaDetect(self.head.Cn, self.strides, self.head.base, pre_nms_top_n, score_thresh, nms_thresh,
                                     max_per_image)
            self._det_key = key
        return self._det(co, bo, im_info)

    def forward_backward(self, image, gt_boxes, im_info, step=0, image_offset=0, step_dev=None, gt_masks=None):
        N, _, H, W = image.shape
        self.plan(N, H, W, gt_boxes.shape[1])
        C = self.backbone.forward(image)
        P = self.neck.forward(C[1:])
        self.head.forward(P)
        loss = self.head.loss_and_grad(gt_boxes, im_info)
        self.head.backward(self.dP)
        lo = 0
        if self._bucket_here(0):
            self._reduce(0, self.mark_head)
            lo = self.mark_head
        self.neck.backward(self.dP, self.dC[1:])
        if self._bucket_here(1):
            self._reduce(lo, self.mark_fpn)
            lo = self.mark_fpn
        for si in (3, 2, 1):
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
            if si == 1:
                self._mark_tail()
            if si == 1 or self._bucket_here(5 - si):
                self._reduce(lo, self.stage_marks[si])
                lo = self.stage_marks[si]
        return (loss,)
