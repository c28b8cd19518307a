"""ResNet-50/101 backbone ("v1b": the stride sits on the 3x3 conv) with frozen BatchNorm folded into the
filters, frozen stem + C2 (forward only), explicit backward through C5..C3.

Plugin slot: models/backbones (/root/reference/README.md:27). MXNet roles replaced: Convolution,
BatchNorm(use_global_stats=True), Activation, Pooling, elemwise_add (README.md:37) - here every bottleneck
conv is ONE kernel with the BN shift, the residual add and the ReLU fused into its epilogue, and every
backward ReLU / shortcut add is fused into a dgrad epilogue.
"""
import os

import torch

from ...ops import dense
from ..utils.layers import ConvLayer, cached_buf


BITMASKS = os.environ.get("MXDET_TUNE_RELU_BITS", "1") == "1"      # 1-bit ReLU masks for the backbone's data gradients
# frozen bottlenecks with 64 mid channels (C2): conv2 -> conv3 as one chained launch (MXDET_TUNE_CHAIN=0: two launches)
CHAIN_FROZEN = os.environ.get("MXDET_TUNE_CHAIN", "1") != "0"
# ... with the NEXT block's conv1 riding along (MXDET_TUNE_CHAIN=1: conv2 -> conv3 only)
CHAIN3_FROZEN = os.environ.get("MXDET_TUNE_CHAIN", "2") not in ("0", "1")


class Bottleneck:
    def __init__(self, name, cin, planes, stride, downsample, trainable, need_dx, arena, ws, device, gen):
        kw = dict(arena=arena, ws=ws, device=device, gen=gen, trainable=trainable, train_bias=False)
        self.trainable, self.need_dx = trainable, need_dx
        # registered in backward completion order: conv3, conv2, conv1 / downsample
        # random-init stand-in for pretrained weights: the residual branch's last conv starts small so that
        # activations stay O(1) through 16 blocks without live BatchNorm statistics (frozen BN is identity here)
        self.conv3 = ConvLayer(name + ".conv3", planes, planes * 4, 1, init_std=0.25 * (2.0 / planes) ** 0.5, **kw)
        self.conv2 = ConvLayer(name + ".conv2", planes, planes, 3, stride, **kw)
        self.conv1 = ConvLayer(name + ".conv1", cin, planes, 1, **kw)
        self.down = ConvLayer(name + ".down", cin, planes * 4, 1, stride, 0, **kw) if downsample else None
        self.x = self.a1 = self.a2 = self.y = self.sc = None
        self.bufs = {}
        self.y_key = "y"      # name of the output buffer (ResNet.forward_front switches the frozen front end's between two)

    def layers(self):
        return [l for l in (self.conv3, self.conv2, self.conv1, self.down) if l is not None]

    def _buf(self, key, shape):
        return cached_buf(self.bufs, key, shape, torch.bfloat16, self.conv1.device)

    def plan(self, x_shape):
        s1 = self.conv1.out_shape(x_shape)
        s2 = self.conv2.out_shape(s1)
        self.conv1.plan(x_shape)
        self.conv2.plan(s1)
        self.conv3.plan(s2)
        if self.down is not None:
            self.down.plan(x_shape)
        return self.conv3.out_shape(s2)

    def _bits(self, key, shape):
        return cached_buf(self.bufs, key, shape[:3] + (shape[3] // 8,), torch.uint8, self.conv1.device)

    def chainable(self):
        """Frozen block with 64 mid channels: conv2 -> conv3 (and the next such block's conv1) run as one launch."""
        return (CHAIN_FROZEN and not self.trainable and self.conv2.stride == 1 and self.conv2.cout == 64 and
                self.conv3.cout == 256)

    def forward(self, x, x_bits=None, a1_ready=False, next_block=None):
        """a1_ready: this block's conv1 output was already written by the previous block's chained launch (self.a1).
        next_block: a following chainable block without projection shortcut -- its conv1 rides along in this block's
        chained launch.
        x_bits: the 1-bit ReLU mask of x (the producer's `bits_out`), or None. Trainable blocks have every convolution
        write the 1-bit mask of its activation next to it: the data gradients read those instead of the activations (a
        forward activation is cold in every cache by the time backward needs it, and the expand-layer data gradients
        are HBM-bound: the mask was a third of their traffic)."""
        self.x, self.x_bits = x, x_bits
        bits = self.trainable and BITMASKS
        a1 = self._buf("a1", self.conv1.out_shape(x.shape))
        s2 = self.conv2.out_shape(a1.shape)
        oshape = self.conv3.out_shape(s2)
        self.a1_bits = self._bits("a1b", a1.shape) if bits else None
        self.a2_bits = self._bits("a2b", s2) if bits else None
        self.y_bits = self._bits("yb", oshape) if bits else None
        if self.down is None:
            if not a1_ready:
                self.a1 = self.conv1.forward(x, relu=True, out=a1, bits_out=self.a1_bits)
            sc = x
        else:       # conv1 and the projection shortcut read the same x: one grouped launch
            sc = self._buf("sc", oshape)
            dense.conv2d_group("fwd", [self.conv1.fwd_call(x, relu=True, out=a1, bits_out=self.a1_bits),
                                       self.down.fwd_call(x, out=sc)], self.conv1.device)
            self.a1 = a1
        if self.chainable():
            # frozen block (C2): nobody reads a2 again -- conv2 and conv3 as ONE launch, the 64-channel map stays in the
            # workgroup (mxdet_conv2d_fwd_chain; bit-identical to the two launches); the next block's conv1 is computed
            # from the block output as it leaves the workgroup
            if dense.PF_TRACE is not None:
                dense.PF_TRACE.append((self.conv2, "f", dense.mem_range(self.conv3.w_bf16, self.conv2.w_bf16), None))
            self.a2 = None
            nb = next_block
            kw = {}
            if nb is not None:
                nb.a1 = nb._buf("a1", nb.conv1.out_shape(oshape))
                kw = dict(w3=nb.conv1.w_bf16, bias3=nb.conv1.bias_f32, relu3=True, out3=nb.a1)
            r = dense.conv2d_forward_chain(self.a1, self.conv2.w_bf16, self.conv2.bias_f32, self.conv3.w_bf16,
                                           self.conv3.bias_f32, sc, relu=True, relu2=True, out=self._buf(self.y_key, oshape),
                                           prefetch=self.conv2.pf_fwd, **kw)
            self.y = r[0] if nb is not None else r
            return self.y
        self.a2 = self.conv2.forward(self.a1, relu=True, out=self._buf("a2", s2), bits_out=self.a2_bits)
        self.y = self.conv3.forward(self.a2, relu=True, residual=sc, out=self._buf(self.y_key, oshape), bits_out=self.y_bits)
        return self.y

    def backward(self, ds, dx_buf, dx_has_grad):
        """ds: gradient w.r.t. the pre-activation sum (already masked by y > 0).

        dx_buf receives d(loss)/d(pre-activation of the producer of x), i.e. the total gradient arriving at x
        masked by (x > 0); if dx_has_grad it already holds a partial gradient (FPN lateral) to add to.
        """
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        c3.backward_weight(self.a2, ds)
        d_a2 = c3.backward_data(ds, self.a2.shape, relu_mask=self.a2, out=self._buf("d_a2", self.a2.shape),
                                relu_bits=self.a2_bits)
        c2.backward_weight(self.a1, d_a2)
        d_a1 = c2.backward_data(d_a2, self.a1.shape, relu_mask=self.a1, out=self._buf("d_a1", self.a1.shape),
                                relu_bits=self.a1_bits)
        c1.backward_weight(self.x, d_a1)
        if self.down is not None:
            self.down.backward_weight(self.x, ds)
        if not self.need_dx:
            return None
        xb = self.x_bits
        if self.down is not None:
            self.down.backward_data(ds, self.x.shape, accumulate=dx_has_grad, out=dx_buf)
            c1.backward_data(d_a1, self.x.shape, residual=dx_buf, relu_mask=self.x, out=dx_buf, relu_bits=xb)
        else:
            if dx_has_grad:
                dense.add_bf16(dx_buf, ds, dx_buf)
                c1.backward_data(d_a1, self.x.shape, residual=dx_buf, relu_mask=self.x, out=dx_buf, relu_bits=xb)
            else:
                c1.backward_data(d_a1, self.x.shape, residual=ds, relu_mask=self.x, out=dx_buf, relu_bits=xb)
        return dx_buf


class ResNet:
    """depth 50: (3,4,6,3), depth 101: (3,4,23,3). Returns C2..C5 (bf16 channels-last)."""

    def __init__(self, depth, arena, ws, device, gen, frozen_stages=1):
        blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
        self.device = device
        # registration order == backward completion order: layer4 ... layer2 (layer1 + stem frozen)
        self.stages = [None] * 4
        cins = [64, 256, 512, 1024]
        for si in (3, 2, 1, 0):
            planes = 64 << si
            trainable = si > frozen_stages - 1 and si >= 1 if frozen_stages >= 1 else True
            stage = []
            for bi in reversed(range(blocks[si])):
                cin = cins[si] if bi == 0 else planes * 4
                stride = 2 if (bi == 0 and si > 0) else 1
                need_dx = trainable and not (bi == 0 and (si == 0 or si - 1 < frozen_stages))
                stage.insert(0, Bottleneck("layer%d.%d" % (si + 1, bi), cin, planes, stride, bi == 0, trainable,
                                           need_dx, arena, ws, device, gen))
            self.stages[si] = stage
        self.stem_w = (torch.randn((64, 7, 7, 3), generator=gen) * (2.0 / 147) ** 0.5).to(torch.bfloat16).to(device)
        self.stem_b = torch.zeros((64,), dtype=torch.float32, device=device)
        self.bufs = {}
        self.outs = None
        self.front_override = None     # see forward()
        self.eager_parity = 0          # which of the two front-end output buffers an eager forward() writes
        self.front_outs = []

    def layers(self):
        return [l for st in self.stages for b in st for l in b.layers()]

    def plan(self, image_shape):
        N, _, H, W = image_shape
        s = (N, ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1, 64)
        shapes = []
        for st in self.stages:
            for b in st:
                s = b.plan(s)
            shapes.append(s)
        return shapes

    def forward(self, image):
        """image: NCHW [N,3,H,W] (f32 or bf16), read directly by the stem kernel. With front_override set (the captured
        step of DetectorBase.capture: the frozen front end ran as a graph of its own) the image is not read: the trainable
        stages start from that tensor."""
        if self.front_override is not None:
            return self.forward_rest(self.front_override)
        return self.forward_rest(self.forward_front(image, self.eager_parity))

    def frozen_front(self):
        """Number of leading stages that are frozen (their output depends on the image only)."""
        n = 0
        for st in self.stages:
            if any(b.trainable for b in st):
                break
            n += 1
        return n

    def forward_front(self, image, parity=0):
        """Stem + max-pool + the frozen stages (C2 with the default frozen_stages = 1): a function of the image alone, so a
        training step may compute it for the NEXT batch while the previous step still runs its weight-gradient tail. The
        result lives in one of two buffers (parity): the previous step's backward still reads the other one."""
        N, _, H, W = image.shape
        H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dev = self.stem_w.device
        # stem + max-pool in one pass: the frozen front end never writes its 2x-resolution map
        x = dense.stem_conv7x7_pool(image, self.stem_w, self.stem_b,
                                    cached_buf(self.bufs, "pool", (N, (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1, 64),
                                               torch.bfloat16, dev))
        nf = self.frozen_front()
        if nf > 0:
            self.stages[nf - 1][-1].y_key = "y%d" % parity if parity else "y"
        self.front_outs = []
        for st in self.stages[:nf]:
            x = self._stage_forward(st, x, None)
            self.front_outs.append(x)
        return x

    def _stage_forward(self, st, x, xb):
        ready = False
        for i, b in enumerate(st):
            nb = st[i + 1] if i + 1 < len(st) else None
            ride = (CHAIN3_FROZEN and nb is not None and b.chainable() and nb.chainable() and nb.down is None and
                    nb.conv1.cout == 64 and nb.conv1.stride == 1)
            x = b.forward(x, xb, a1_ready=ready, next_block=nb if ride else None)
            ready = ride
            xb = b.y_bits
        self._xb = xb
        return x

    def forward_rest(self, x):
        nf = self.frozen_front()
        if self.front_override is None:
            outs = list(self.front_outs[:nf])
        else:                               # only the last frozen stage's output exists as a tensor of this step
            outs = ([None] * (nf - 1) + [x]) if nf else []
        xb = None                       # 1-bit ReLU mask of x, when its producer is a trainable block
        for st in self.stages[nf:]:
            x = self._stage_forward(st, x, xb)
            xb = self._xb
            outs.append(x)
        self.outs = outs
        return outs

    def backward(self, dC):
        """dC[i]: bf16 buffer holding the FPN-lateral gradient w.r.t. C(i+2) (un-masked) for i = 1..3 (C3..C5);
        dC[3] (C5) must already be masked by (C5 > 0). Walks layer4 -> layer2."""
        for si in (3, 2, 1):
            stage = self.stages[si]
            if not stage[0].trainable:
                break
            ds = dC[si]
            for bi in reversed(range(len(stage))):
                b = stage[bi]
                if bi > 0:
                    dxb = b._buf("dx", b.x.shape)
                    ds = b.backward(ds, dxb, False)
                else:
                    # input of the stage's first block is the previous stage's output: add to its lateral gradient
                    if b.need_dx:
                        ds = b.backward(ds, dC[si - 1], True)
                    else:
                        b.backward(ds, None, False)
