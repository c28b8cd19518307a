"""Feature Pyramid Network neck: 1x1 laterals, top-down nearest-2x add, 3x3 outputs, P6 = stride-2 subsample.

Plugin slot: models/necks (/root/reference/README.md:31). MXNet roles replaced: Convolution, UpSampling(nearest),
elemwise_add, Pooling (README.md:37). The top-down add is fused into the lateral conv's epilogue (the coarser map is
read nearest-neighbour as the residual), so the merged map is written once and never re-read for the add.
"""
import torch

from ...ops import dense
from ..utils.layers import ConvLayer, cached_buf


class FPN:
    def __init__(self, in_channels, out_channels, arena, ws, device, gen, extra_p6=True):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        L = len(in_channels)
        # registered in backward completion order: output convs fine -> coarse, then laterals
        # Xavier-style init (no ReLU follows these convs, so He-normal would double the variance per layer)
        self.outs = [ConvLayer("fpn.out%d" % (i + 2), out_channels, out_channels, 3,
                               init_std=(1.0 / (9 * out_channels)) ** 0.5, **kw) for i in range(L)]
        self.lats = [ConvLayer("fpn.lat%d" % (i + 2), in_channels[i], out_channels, 1,
                               init_std=(1.0 / in_channels[i]) ** 0.5, **kw) for i in range(L)]
        self.L, self.C, self.extra_p6 = L, out_channels, extra_p6
        self.device = device
        self.bufs = {}
        self.inner = self.feats = self.P = None

    def layers(self):
        return self.outs + self.lats

    def plan(self, c_shapes):
        for i, s in enumerate(c_shapes):
            self.lats[i].plan(s)
            self.outs[i].plan((s[0], s[1], s[2], self.C))

    def _buf(self, key, shape):
        return cached_buf(self.bufs, key, shape, torch.bfloat16, self.device)

    def forward(self, feats):
        L = self.L
        self.feats = feats
        inner = [None] * L
        for i in reversed(range(L)):
            shape = feats[i].shape[:3] + (self.C,)
            res = inner[i + 1] if i + 1 < L else None
            inner[i] = self.lats[i].forward(feats[i], residual=res, res_upsample=res is not None,
                                            out=self._buf("inner%d" % i, shape))
        P = [self._buf("P%d" % i, inner[i].shape) for i in range(L)]
        self.outs[0].forward(inner[0], out=P[0])          # the finest level fills the chip on its own
        dense.conv2d_group("fwd", [self.outs[i].fwd_call(inner[i], out=P[i]) for i in range(1, L)], self.device)
        if self.extra_p6:
            N, H, W, Cc = P[-1].shape
            P.append(dense.subsample2(P[-1], self._buf("P6", (N, (H + 1) // 2, (W + 1) // 2, Cc))))
        self.inner, self.P = inner, P
        return P

    def backward(self, dP, dC, c_needs_grad):
        """dP[i]: gradient w.r.t. P(i+2) (bf16); dC[i]: output buffers for the gradient w.r.t. C(i+2);
        c_needs_grad[i] says whether the backbone wants it. The top level's dC is masked by (C > 0) here
        (the lateral is its only consumer); the others stay un-masked for the backbone to finish."""
        L = self.L
        if self.extra_p6:
            dense.subsample2_backward(dP[L], dP[L - 1], accumulate=True)
        dinner = [self._buf("dinner%d" % i, self.inner[i].shape) for i in range(L)]
        for i in range(L):
            self.outs[i].backward_weight(self.inner[i], dP[i])
        self.outs[0].backward_data(dP[0], self.inner[0].shape, out=dinner[0])
        dense.conv2d_group("dgrad", [self.outs[i].dgrad_call(dP[i], self.inner[i].shape, out=dinner[i])
                                     for i in range(1, L)], self.device)
        for i in range(1, L):                    # top-down path backwards: fine -> coarse, in order
            dense.upsample2_backward(dinner[i - 1], dinner[i], accumulate=True)
        calls = []
        for i in range(L):
            self.lats[i].backward_weight(self.feats[i], dinner[i])
            if c_needs_grad[i]:
                top = i == L - 1
                calls.append(self.lats[i].dgrad_call(dinner[i], self.feats[i].shape,
                                                     relu_mask=self.feats[i] if top else None, out=dC[i]))
        if calls:
            dense.conv2d_group("dgrad", calls, self.device)

class RetinaFPN:
    """RetinaNet pyramid P3..P7: laterals + top-down on C3..C5, P6 = 3x3/2 conv on C5, P7 = 3x3/2 conv on ReLU(P6)
    (RetinaNet variant)."""

    def __init__(self, in_channels, out_channels, arena, ws, device, gen):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        L = len(in_channels)   # C3, C4, C5
        xav = lambda cin, k: (1.0 / (k * k * cin)) ** 0.5   # noqa: E731
        self.p7 = ConvLayer("fpn.p7", out_channels, out_channels, 3, 2, init_std=xav(out_channels, 3), **kw)
        self.outs = [ConvLayer("fpn.out%d" % (i + 3), out_channels, out_channels, 3, init_std=xav(out_channels, 3), **kw)
                     for i in range(L)]
        self.p6 = ConvLayer("fpn.p6", in_channels[-1], out_channels, 3, 2, init_std=xav(in_channels[-1], 3), **kw)
        self.lats = [ConvLayer("fpn.lat%d" % (i + 3), in_channels[i], out_channels, 1, init_std=xav(in_channels[i], 1), **kw)
                     for i in range(L)]
        self.L, self.C, self.device = L, out_channels, device
        self.bufs = {}

    def layers(self):
        return [self.p7] + self.outs + [self.p6] + self.lats

    def _buf(self, key, shape):
        return cached_buf(self.bufs, key, shape, torch.bfloat16, self.device)

    def plan(self, c_shapes):
        for i, s in enumerate(c_shapes):
            self.lats[i].plan(s)
            self.outs[i].plan((s[0], s[1], s[2], self.C))
        self.p6.plan(c_shapes[-1])
        self.p7.plan(self.p6.out_shape(c_shapes[-1]))
        s6 = self.p6.out_shape(c_shapes[-1])
        return [(s[0], s[1], s[2], self.C) for s in c_shapes] + [s6, self.p7.out_shape(s6)]

    def forward(self, feats):
        L = self.L
        self.feats = feats
        inner = [None] * L
        for i in reversed(range(L)):
            res = inner[i + 1] if i + 1 < L else None
            inner[i] = self.lats[i].forward(feats[i], residual=res, res_upsample=res is not None,
                                            out=self._buf("inner%d" % i, feats[i].shape[:3] + (self.C,)))
        P = [self.outs[i].forward(inner[i], out=self._buf("P%d" % i, inner[i].shape)) for i in range(L)]
        self.p6_raw = self.p6.forward(feats[-1], out=self._buf("P6", self.p6.out_shape(feats[-1].shape)))
        self.p6_relu = dense.relu_forward(self.p6_raw, self._buf("P6r", self.p6_raw.shape))
        p7 = self.p7.forward(self.p6_relu, out=self._buf("P7", self.p7.out_shape(self.p6_relu.shape)))
        self.inner, self.P = inner, P + [self.p6_raw, p7]
        return self.P

    def backward(self, dP, dC):
        """dP[0..4] = gradients w.r.t. P3..P7; dC[0..2] receive gradients w.r.t. C3..C5 (C5 masked by C5 > 0)."""
        L = self.L
        # P7 = conv(ReLU(P6)): d(P6) = dP6 (heads) + (P6 > 0) * dgrad_p7(dP7)
        self.p7.backward_weight(self.p6_relu, dP[L + 1])
        t = self.p7.backward_data(dP[L + 1], self.p6_relu.shape, relu_mask=self.p6_relu, out=self._buf("dP6r", self.p6_relu.shape))
        dense.add_bf16(dP[L], t, dP[L])
        dinner = [None] * L
        for i in range(L):
            self.outs[i].backward_weight(self.inner[i], dP[i])
            dinner[i] = self.outs[i].backward_data(dP[i], self.inner[i].shape, out=self._buf("dinner%d" % i, self.inner[i].shape))
            if i > 0:
                dense.upsample2_backward(dinner[i - 1], dinner[i], accumulate=True)
        self.p6.backward_weight(self.feats[-1], dP[L])
        self.p6.backward_data(dP[L], self.feats[-1].shape, out=dC[L - 1])
        for i in range(L):
            self.lats[i].backward_weight(self.feats[i], dinner[i])
            top = i == L - 1
            self.lats[i].backward_data(dinner[i], self.feats[i].shape, residual=dC[i] if top else None,
                                       relu_mask=self.feats[i] if top else None, out=dC[i])
