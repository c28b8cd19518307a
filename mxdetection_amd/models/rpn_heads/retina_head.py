"""RetinaNet dense head: two 4-conv towers (class / box) shared across P3..P7, 9 anchors per cell, sigmoid focal loss
and smooth-L1 box loss on every anchor (no sampling, no proposals).

Declared slot: the reference only names `rpn_heads` / `bbox_heads` for dense heads (/root/reference/README.md:28-29);
RetinaNet is BASELINE.json config 5. Class logits are padded 9*80 = 720 -> 768 channels and deltas 36 -> 64 so that
dgrad's reduction dim is a multiple of 64.
"""
import math

import torch

from ...core import anchor as A_
from ...core import loss as L_
from ..utils.layers import ConvLayer


class RetinaHead:
    def __init__(self, channels, strides, arena, ws, device, gen, num_classes=80, num_convs=4, ratios=(0.5, 1.0, 2.0),
                 octave_scales=(1.0, 2.0 ** (1.0 / 3.0), 2.0 ** (2.0 / 3.0)), anchor_scale=4.0, fg_thresh=0.5,
                 bg_thresh=0.4, alpha=0.25, gamma=2.0, sigma=3.0, prior=0.01):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        self.A, self.Cn = len(ratios) * len(octave_scales), num_classes
        self.ld_cls = (self.A * num_classes + 63) // 64 * 64
        self.ld_reg = (self.A * 4 + 63) // 64 * 64
        # registration = backward completion order
        self.cls_out = ConvLayer("retina.cls_out", channels, self.ld_cls, 3, init_std=0.01, **kw)
        self.box_out = ConvLayer("retina.box_out", channels, self.ld_reg, 3, init_std=0.01, **kw)
        self.cls_convs = [ConvLayer("retina.cls%d" % i, channels, channels, 3, init_std=0.01, **kw)
                          for i in reversed(range(num_convs))][::-1]
        self.box_convs = [ConvLayer("retina.box%d" % i, channels, channels, 3, init_std=0.01, **kw)
                          for i in reversed(range(num_convs))][::-1]
        self.prior_bias = -math.log((1.0 - prior) / prior)
        self.strides = list(strides)
        scales = [anchor_scale * o for o in octave_scales]
        self.base = [torch.from_numpy(A_.generate_base_anchors(s, ratios, scales)).to(device) for s in strides]
        self.fg_thresh, self.bg_thresh, self.alpha, self.gamma, self.sigma = fg_thresh, bg_thresh, alpha, gamma, sigma
        self.device, self.C = device, channels
        self.bufs = {}

    def layers(self):
        return [self.cls_out, self.box_out] + list(reversed(self.cls_convs)) + list(reversed(self.box_convs))

    def post_materialize(self):
        b = self.cls_out.bias_f32
        b.zero_()
        b[: self.A * self.Cn] = self.prior_bias      # focal-loss prior: every anchor starts at p = 0.01

    def _buf(self, key, shape, dtype=torch.bfloat16, zero=False):
        b = self.bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape):
            b = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self.bufs[key] = b
        return b

    def plan(self, p_shapes, g_max):
        for s in p_shapes:
            for c in self.cls_convs + self.box_convs + [self.cls_out, self.box_out]:
                c.plan(s)
        self.level_shapes = [(s[1], s[2]) for s in p_shapes]
        N = p_shapes[0][0]
        self.anchors = torch.cat([A_.generate_anchors(self.base[l], H, W, self.strides[l])
                                  for l, (H, W) in enumerate(self.level_shapes)])
        self.level_offsets = [0]
        for (H, W) in self.level_shapes:
            self.level_offsets.append(self.level_offsets[-1] + H * W * self.A)
        At = self.anchors.shape[0]
        dev = self.device
        self.at_ws = A_.AnchorTargetWorkspace(N, At, g_max, dev)
        self.at_out = (torch.empty((N, At), dtype=torch.int32, device=dev), torch.empty((N, At), dtype=torch.int32, device=dev),
                       torch.empty((N, At, 4), dtype=torch.float32, device=dev), torch.empty((N, At), dtype=torch.float32, device=dev))
        self.cls_labels = torch.empty((N, At), dtype=torch.int32, device=dev)
        self.num_fg = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.nparts = [L_.retina_loss_num_partials(N, H, W, self.A) for (H, W) in self.level_shapes]
        self.partial = torch.zeros((2 * sum(self.nparts),), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((2,), dtype=torch.float32, device=dev)

    def forward(self, P):
        self.P = P
        self.cact, self.bact, self.co, self.bo = [], [], [], []
        for l, p in enumerate(P):
            x, acts = p, [p]
            for i, c in enumerate(self.cls_convs):
                x = c.forward(x, relu=True, out=self._buf("c%d_%d" % (l, i), p.shape))
                acts.append(x)
            self.cact.append(acts)
            self.co.append(self.cls_out.forward(x, out=self._buf("co%d" % l, p.shape[:3] + (self.ld_cls,))))
            x, acts = p, [p]
            for i, c in enumerate(self.box_convs):
                x = c.forward(x, relu=True, out=self._buf("b%d_%d" % (l, i), p.shape))
                acts.append(x)
            self.bact.append(acts)
            self.bo.append(self.box_out.forward(x, out=self._buf("bo%d" % l, p.shape[:3] + (self.ld_reg,))))
        return self.co, self.bo

    def loss_and_grad(self, gt_boxes, im_info, loss_scale=1.0):
        labels, matched, targets, _ = A_.assign_anchor(self.anchors, gt_boxes, im_info, self.fg_thresh, self.bg_thresh,
                                                       1.0e6, 0, 0.5, 0, 0, 0, self.at_ws, self.at_out)
        L_.anchor_class_labels(labels, matched, gt_boxes, self.cls_labels, self.num_fg)
        self.gco, self.gbo = [], []
        off = 0
        for l in range(len(self.co)):
            gc = self._buf("gco%d" % l, self.co[l].shape, zero=True)     # padding channels stay zero
            gb = self._buf("gbo%d" % l, self.bo[l].shape, zero=True)
            L_.retina_loss_level(self.co[l], self.bo[l], self.A, self.Cn, self.cls_labels, targets, self.level_offsets[l],
                                 self.alpha, self.gamma, self.sigma, self.num_fg, loss_scale, gc, gb, self.partial[2 * off:])
            off += self.nparts[l]
            self.gco.append(gc)
            self.gbo.append(gb)
        L_.loss_finalize(self.partial, off, 2, self.loss)
        return self.loss

    def backward(self, dP):
        """Writes d(loss)/d(P_l) into dP[l] (overwrites)."""
        n = len(self.cls_convs)
        for l in range(len(self.co)):
            acc = l > 0
            for out_layer, convs, acts, g, key, first in ((self.cls_out, self.cls_convs, self.cact[l], self.gco[l], "c", True),
                                                          (self.box_out, self.box_convs, self.bact[l], self.gbo[l], "b", False)):
                x = acts[-1]
                out_layer.backward_weight(x, g, accumulate=acc)
                d = out_layer.backward_data(g, x.shape, relu_mask=x, out=self._buf("d%s%d_%d" % (key, l, n), x.shape))
                for i in reversed(range(n)):
                    xin = acts[i]
                    convs[i].backward_weight(xin, d, accumulate=acc)
                    if i > 0:
                        d = convs[i].backward_data(d, xin.shape, relu_mask=xin, out=self._buf("d%s%d_%d" % (key, l, i), xin.shape))
                    else:   # into the pyramid gradient: class tower writes, box tower adds
                        convs[i].backward_data(d, xin.shape, accumulate=not first, out=dP[l])
