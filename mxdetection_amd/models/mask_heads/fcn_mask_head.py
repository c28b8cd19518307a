"""Mask R-CNN FCN mask head: RoIAlign 14x14 on the foreground rois -> 4 x (conv3x3 256 + ReLU) -> deconv 2x2/2 256 +
ReLU -> conv1x1 -> per-class 28x28 logits; targets from core/mask; per-pixel sigmoid BCE on the GT class channel.

Plugin slot: models/mask_heads (/root/reference/README.md:30) with core/mask (README.md:18). The deconvolution runs
as a 1x1 convolution 256 -> 4*256 on the MFMA kernel (bias + ReLU fused) followed by a pixel shuffle; the class
logits are padded 80 -> 128 channels so that dgrad's reduction dim is a multiple of 64.
"""
import torch

from ...core import mask as M_
from ...ops import dense
from ..utils.layers import ConvLayer, cached_buf


class FCNMaskHead:
    def __init__(self, channels, arena, ws, device, gen, num_classes=81, num_convs=4, size=28, rois_per_image=128):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen)
        self.nc, self.S, self.Rimg, self.device, self.C = num_classes, size, rois_per_image, device, channels
        self.cpad = (num_classes - 1 + 63) // 64 * 64
        # registration = backward completion order
        self.logits = ConvLayer("mask.logits", channels, self.cpad, 1, init_std=0.001, cout_real=num_classes - 1, **kw)
        self.deconv = ConvLayer("mask.deconv", channels, 4 * channels, 1, **kw)   # 2x2/2 deconv as 1x1 conv + shuffle
        self.convs = [ConvLayer("mask.conv%d" % i, channels, channels, 3, **kw) for i in reversed(range(num_convs))][::-1]
        self.bufs = {}

    def layers(self):
        return [self.logits, self.deconv] + list(reversed(self.convs))

    def _buf(self, key, shape, dtype=torch.bfloat16, zero=False):
        return cached_buf(self.bufs, key, shape, dtype, self.device, zero)

    def plan(self, N):
        R = N * self.Rimg
        h = self.S // 2
        for c in self.convs:
            c.plan((R, h, h, self.C))
        self.deconv.plan((R, h, h, self.C))
        self.logits.plan((R, self.S, self.S, self.C))
        self.loss = torch.zeros((1,), dtype=torch.float32, device=self.device)
        self.loss_ws = M_.mask_loss_workspace(R, self.S, self.device)

    def select_rois(self, bbox_head):
        """The box head writes its sampled rois foreground first, so the first `Rimg` slots of every image hold all
        foreground rois (<= 25% of 512 = 128) followed by background / padding, which the targets mark as ignored."""
        N = bbox_head.rois.shape[0]
        k = self.Rimg
        self.rois = bbox_head.rois[:, :k].reshape(N * k, 5).contiguous()
        self.matched = bbox_head.matched[:, :k].reshape(-1).contiguous()
        self.roi_labels = bbox_head.labels[:, :k].reshape(-1).contiguous()
        return self.rois

    def targets(self, gt_masks):
        R = self.rois.shape[0]
        out = (self._buf("tg", (R, self.S, self.S), torch.uint8), self._buf("cls", (R,), torch.int32))
        self.tg, self.cls = M_.mask_target(self.rois, self.matched, self.roi_labels, gt_masks, self.S, out)

    def forward(self, pooled):
        """pooled bf16 [R,14,14,C]"""
        x = pooled
        self.acts = [x]
        for i, c in enumerate(self.convs):
            x = c.forward(x, relu=True, out=self._buf("a%d" % i, x.shape))
            self.acts.append(x)
        R, h, w, _ = x.shape
        self.d4 = self.deconv.forward(x, relu=True, out=self._buf("d4", (R, h, w, 4 * self.C)))
        self.up = dense.pixel_shuffle2(self.d4, self._buf("up", (R, 2 * h, 2 * w, self.C)))
        self.o = self.logits.forward(self.up, out=self._buf("o", (R, 2 * h, 2 * w, self.cpad)))
        return self.o

    def loss_and_grad(self, loss_scale=1.0):
        self.go = self._buf("go", self.o.shape)
        M_.mask_loss(self.o, self.cls, self.tg, self.loss, self.go, self.loss_ws, loss_scale)
        return self.loss

    def backward(self):
        """Returns d(loss)/d(pooled) bf16 [R,14,14,C]."""
        self.logits.backward_weight(self.up, self.go)
        d_up = self.logits.backward_data(self.go, self.up.shape, out=self._buf("d_up", self.up.shape))
        d_d4 = dense.pixel_shuffle2_inv_relu(d_up, self.d4, self._buf("d_d4", self.d4.shape))
        x = self.acts[-1]
        self.deconv.backward_weight(x, d_d4)
        g = self.deconv.backward_data(d_d4, x.shape, relu_mask=x, out=self._buf("g%d" % len(self.convs), x.shape))
        for i in reversed(range(len(self.convs))):
            xin = self.acts[i]
            self.convs[i].backward_weight(xin, g)
            g = self.convs[i].backward_data(g, xin.shape, relu_mask=xin if i > 0 else None,
                                            out=self._buf("g%d" % i, xin.shape))
        return g
