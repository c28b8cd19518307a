"""Two-FC box head: flatten(7x7x256) -> FC 1024 + ReLU -> FC 1024 + ReLU -> fused (cls 81 | reg 324) FC.

Plugin slot: models/bbox_heads (/root/reference/README.md:29) with core/bbox + core/loss (README.md:17,19).
The FCs run on the same MFMA implicit-GEMM kernels as the convolutions (1x1 on [R,1,1,C] tensors); cls and reg
share one GEMM whose output is padded to 448 columns (a multiple of 64, so dgrad can reduce over it).
"""
import os

import torch

from ...core import bbox as B_
from ...core import loss as L_
from ...ops import dense
from ..utils.layers import ConvLayer, cached_buf


class BBoxHead:
    def __init__(self, in_features, arena, ws, device, gen, num_classes=81, fc_dim=1024, rois_per_image=512,
                 fg_fraction=0.25, fg_thresh=0.5, bg_hi=0.5, bg_lo=0.0, stds=(0.1, 0.1, 0.2, 0.2), sigma=1.0,
                 seed=99):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        self.nc = num_classes
        self.reg_dim = 4 * num_classes
        self.ld = (num_classes + self.reg_dim + 63) // 64 * 64
        self.fc_out = ConvLayer("bbox.fc_out", fc_dim, self.ld, 1, init_std=0.01, cout_real=num_classes + self.reg_dim, **kw)
        self.fc2 = ConvLayer("bbox.fc2", fc_dim, fc_dim, 1, **kw)
        self.fc1 = ConvLayer("bbox.fc1", in_features, fc_dim, 1, **kw)
        # checkpoint layout (DetectorBase._to_mx): fully connected layers are stored 2-D; fc1's input is the pooled
        # [7,7,C] block flattened (H, W, C) here and (C, H, W) in an MXNet FullyConnected after a Flatten of NCHW
        pooled = 7
        self.fc1.fc_in_hwc = (pooled, pooled, in_features // (pooled * pooled)) if in_features % (pooled * pooled) == 0 else ()
        self.fc2.fc_in_hwc = ()
        self.fc_out.fc_in_hwc = ()
        self.R, self.fg_fraction, self.fg_thresh, self.bg_hi, self.bg_lo = rois_per_image, fg_fraction, fg_thresh, bg_hi, bg_lo
        self.stds, self.sigma, self.seed, self.device = stds, sigma, seed, device
        self.in_features, self.fc_dim = in_features, fc_dim
        self.fc1_ksplit = int(os.environ.get("MXDET_TUNE_FC1_KSPLIT", "4"))
        self.bufs = {}

    def layers(self):
        return [self.fc_out, self.fc2, self.fc1]

    def _buf(self, key, shape, dtype=torch.bfloat16, zero=False):
        return cached_buf(self.bufs, key, shape, dtype, self.device, zero)

    def plan(self, N):
        R = N * self.R
        self.fc1.plan((R, 1, 1, self.in_features))
        self.fc2.plan((R, 1, 1, self.fc_dim))
        self.fc_out.plan((R, 1, 1, self.fc_dim))
        self.loss = torch.zeros((2,), dtype=torch.float32, device=self.device)
        self.loss_ws = L_.loss_workspace(R, self.device)

    def sample(self, rois, num_rois, gt_boxes, step, image_offset, step_dev=None):
        """proposal-target: returns rois [N*R,5] and keeps labels / targets / weights for the loss."""
        out = B_.sample_rois(rois, num_rois, gt_boxes, self.R, self.fg_fraction, self.fg_thresh, self.bg_hi, self.bg_lo,
                             self.nc, False, (0.0, 0.0, 0.0, 0.0), self.stds, self.seed, step, image_offset, step_dev)
        self.rois, self.labels, self.tgt, self.wgt, self.matched, self.num_fg = out
        return self.rois.view(-1, 5)

    def forward(self, pooled):
        R = pooled.shape[0]
        self.x = pooled.view(R, 1, 1, -1)
        # fc1 is 196 K-steps on 16 x 16 tiles of 64 x 64: one latency-bound wave per SIMD on its own (118 us in the step
        # while nothing else runs). Split four ways the grid fills the chip; the fold is deterministic (split order).
        ks = self.fc1_ksplit if (R * self.fc_dim) % (64 * 64 * 8) == 0 and R % 64 == 0 else 1
        if ks > 1:
            need = 4 * ks * R * self.fc_dim
            ws = self._buf("fc1_ws", (need,), dtype=torch.uint8)
            self.h1 = dense.conv2d_forward_splitk(self.x, self.fc1.w_bf16, self.fc1.bias_f32, None, True, ks,
                                                  self._buf("h1", (R, 1, 1, self.fc_dim)), ws)
        else:
            self.h1 = self.fc1.forward(self.x, relu=True, out=self._buf("h1", (R, 1, 1, self.fc_dim)))
        self.h2 = self.fc2.forward(self.h1, relu=True, out=self._buf("h2", (R, 1, 1, self.fc_dim)))
        self.o = self.fc_out.forward(self.h2, out=self._buf("o", (R, 1, 1, self.ld)))
        return self.o

    def loss_and_grad(self, loss_scale=1.0):
        R = self.o.shape[0]
        o2 = self.o.view(R, self.ld)
        self.go = self._buf("go", (R, 1, 1, self.ld), zero=True)   # padding columns stay zero forever
        g2 = self.go.view(R, self.ld)
        L_.rcnn_loss(o2, o2[:, self.nc:], self.labels, self.tgt, self.wgt, self.nc, self.reg_dim, self.ld, self.ld,
                     self.sigma, 1.0 / R, loss_scale, g2, g2[:, self.nc:], self.loss, self.loss_ws)
        return self.loss

    def backward(self):
        R = self.o.shape[0]
        self.fc_out.backward_weight(self.h2, self.go)
        d_h2 = self.fc_out.backward_data(self.go, self.h2.shape, relu_mask=self.h2, out=self._buf("dh2", self.h2.shape))
        self.fc2.backward_weight(self.h1, d_h2)
        d_h1 = self.fc2.backward_data(d_h2, self.h1.shape, relu_mask=self.h1, out=self._buf("dh1", self.h1.shape))
        self.fc1.backward_weight(self.x, d_h1)
        d_x = self.fc1.backward_data(d_h1, self.x.shape, out=self._buf("dx", self.x.shape))
        return d_x
