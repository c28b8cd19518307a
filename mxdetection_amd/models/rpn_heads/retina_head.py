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
from ...ops import dense
from ..utils.layers import ConvLayer, cached_buf


class RetinaHead:
    def __init__(self, channels, strides, arena, ws, device, gen, num_classes=80, num_convs=4, ratios=(0.5, 1.0, 2.0),
                 octave_scales=(1.0, 2.0 ** (1.0 / 3.0), 2.0 ** (2.0 / 3.0)), anchor_scale=4.0, fg_thresh=0.5,
                 bg_thresh=0.4, alpha=0.25, gamma=2.0, sigma=3.0, prior=0.01):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        self.A, self.Cn = len(ratios) * len(octave_scales), num_classes
        self.ld_cls = (self.A * num_classes + 63) // 64 * 64
        self.ld_reg = (self.A * 4 + 63) // 64 * 64
        # registration = backward completion order
        self.cls_out = ConvLayer("retina.cls_out", channels, self.ld_cls, 3, init_std=0.01, cout_real=self.A * num_classes, **kw)
        self.box_out = ConvLayer("retina.box_out", channels, self.ld_reg, 3, init_std=0.01, cout_real=self.A * 4, **kw)
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
        return cached_buf(self.bufs, key, shape, dtype, self.device, zero)

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
        """Layer i of BOTH towers on ALL levels is one grouped launch (10 independent convolutions): 5 launches for the
        whole head instead of 50; the small levels ride in the shadow of P3."""
        self.P = P
        L, n = len(P), len(self.cls_convs)
        self.cact = [[p] for p in P]
        self.bact = [[p] for p in P]
        for i in range(n):
            calls = []
            for l, p in enumerate(P):
                co = self._buf("c%d_%d" % (l, i), p.shape)
                bo = self._buf("b%d_%d" % (l, i), p.shape)
                calls.append(self.cls_convs[i].fwd_call(self.cact[l][-1], relu=True, out=co))
                calls.append(self.box_convs[i].fwd_call(self.bact[l][-1], relu=True, out=bo))
                self.cact[l].append(co)
                self.bact[l].append(bo)
            dense.conv2d_group("fwd", calls, self.device)
        self.co = [self._buf("co%d" % l, p.shape[:3] + (self.ld_cls,)) for l, p in enumerate(P)]
        self.bo = [self._buf("bo%d" % l, p.shape[:3] + (self.ld_reg,)) for l, p in enumerate(P)]
        dense.conv2d_group("fwd", [self.cls_out.fwd_call(self.cact[l][-1], out=self.co[l]) for l in range(L)] +
                           [self.box_out.fwd_call(self.bact[l][-1], out=self.bo[l]) for l in range(L)], self.device)
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
        """Writes d(loss)/d(P_l) into dP[l] (overwrites). Same grouping as forward; weight gradients of a filter are
        summed over the levels (by the grouped plan when the workspace groups, by accumulate flags otherwise)."""
        n, L = len(self.cls_convs), len(self.co)
        grouped = self.cls_out.ws.grouping
        dbuf = lambda key, l, i, shape: self._buf("d%s%d_%d" % (key, l, i), shape)   # noqa: E731
        dc = [None] * L
        db = [None] * L
        calls = []
        for l in range(L):
            xc, xb = self.cact[l][-1], self.bact[l][-1]
            self.cls_out.backward_weight(xc, self.gco[l], accumulate=(l > 0) and not grouped)
            self.box_out.backward_weight(xb, self.gbo[l], accumulate=(l > 0) and not grouped)
            dc[l], db[l] = dbuf("c", l, n, xc.shape), dbuf("b", l, n, xb.shape)
            calls.append(self.cls_out.dgrad_call(self.gco[l], xc.shape, relu_mask=xc, out=dc[l]))
            calls.append(self.box_out.dgrad_call(self.gbo[l], xb.shape, relu_mask=xb, out=db[l]))
        dense.conv2d_group("dgrad", calls, self.device)
        for i in reversed(range(n)):
            calls_c, calls_b = [], []
            for l in range(L):
                xc, xb = self.cact[l][i], self.bact[l][i]
                self.cls_convs[i].backward_weight(xc, dc[l], accumulate=(l > 0) and not grouped)
                self.box_convs[i].backward_weight(xb, db[l], accumulate=(l > 0) and not grouped)
                if i > 0:
                    nc, nb = dbuf("c", l, i, xc.shape), dbuf("b", l, i, xb.shape)
                    calls_c.append(self.cls_convs[i].dgrad_call(dc[l], xc.shape, relu_mask=xc, out=nc))
                    calls_b.append(self.box_convs[i].dgrad_call(db[l], xb.shape, relu_mask=xb, out=nb))
                    dc[l], db[l] = nc, nb
                else:   # into the pyramid gradient: the class tower writes, then the box tower adds
                    calls_c.append(self.cls_convs[i].dgrad_call(dc[l], xc.shape, out=dP[l]))
                    calls_b.append(self.box_convs[i].dgrad_call(db[l], xb.shape, accumulate=True, out=dP[l]))
            if i > 0:
                dense.conv2d_group("dgrad", calls_c + calls_b, self.device)
            else:
                dense.conv2d_group("dgrad", calls_c, self.device)
                dense.conv2d_group("dgrad", calls_b, self.device)
