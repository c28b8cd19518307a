"""Parameter arenas and the convolution layer used by every model part.

Design (MI355X-first, not MXNet's per-NDArray executor):
  * all trainable parameters live in ONE flat fp32 arena (master weights), with a parallel flat fp32
    gradient arena, momentum arena and bf16 working copy. The optimizer is one launch over the arena and
    data-parallel all-reduce works on contiguous slices of the gradient arena (buckets) - no per-tensor
    push/pull as in MXNet's kvstore (/root/reference/README.md:37).
  * parameters are registered in *backward completion order* (box head first, C3 last), so a bucket is a
    contiguous arena slice that becomes final while backward is still running.
  * frozen BatchNorm (use_global_stats) is folded into the filters and a per-channel bias once at
    construction; frozen layers (stem, C2) keep bf16 filters only.
"""
import contextlib
import os
import math

import torch

from ...ops import dense


def cached_buf(bufs, key, shape, dtype, device, zero=False):
    """Activation / gradient buffer cache of a model part, keyed by (name, shape, dtype): a call at another shape
    (inference with 1000 rois per image after training with 512) gets buffers of its own and never frees the ones a
    captured hipGraph holds by address."""
    k = (key, tuple(shape), dtype)
    b = bufs.get(k)
    if b is None:
        b = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=device)
        bufs[k] = b
    return b


class Workspace:
    """One caller-owned scratch buffer shared by every wgrad call (the C-ABI never allocates).

    `side` (optional HIP stream): weight-gradient kernels only feed the optimizer, so they are issued on a second
    stream that forks from the main stream at each call (after the producer of dy) and is joined before the
    gradient exchange / optimizer. They then overlap the dgrad chain instead of sitting on its critical path."""

    def __init__(self, device):
        self.device = device
        self.need = 0
        self.buf = None
        self._retired = []
        self.side = None
        # grouped mode: backward_weight() calls are recorded and issued together at flush() (one launch pair per
        # group -- a ResNet stage, the FPN, a head -- instead of two launches per layer)
        self.grouping = False
        self.pending = []
        self.plans = {}
        self.gbuf = None

    def defer(self, layer, x, dy):
        self.pending.append((layer, x, dy))

    def flush(self):
        """Issue the recorded weight gradients (on the side stream if there is one)."""
        if not self.pending:
            return
        items, self.pending = self.pending, []
        if "wgrad" in os.environ.get("MXDET_ABL_SKIP", ""):       # timing-only ablation, see detector._ABL
            return
        key = tuple((id(l), x.data_ptr(), dy.data_ptr()) for l, x, dy in items)
        plan = self.plans.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if plan is None and not capturing:
            calls = [(x, dy, l.k, l.k, l.stride, l.pad, l.arena.view(l.wi, "g"),
                      l.arena.view(l.bi, "g") if l.train_bias else None, False) for l, x, dy in items]
            plan = dense.GroupedWgrad(calls, self.device)
            self.plans[key] = plan
            if self.gbuf is None or self.gbuf.numel() < plan.workspace_bytes:
                self._retired.append(self.gbuf)
                self.gbuf = torch.empty((max(plan.workspace_bytes, 256),), dtype=torch.uint8, device=self.device)
        ctx = self.fork()
        with (ctx if ctx is not None else contextlib.nullcontext()):
            if plan is None or self.gbuf is None or self.gbuf.numel() < plan.workspace_bytes:
                # first seen under capture (no eager warm-up): a table cannot be uploaded now -- per-layer launches
                for l, x, dy in items:
                    dense.conv2d_wgrad(x, dy, l.k, l.k, l.stride, l.pad, l.arena.view(l.wi, "g"),
                                       l.arena.view(l.bi, "g") if l.train_bias else None, False, self.get())
            else:
                plan.launch(self.gbuf)
    def fork(self):
        if self.side is None:
            return None
        self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def require(self, nbytes):
        self.need = max(self.need, int(nbytes))

    def get(self):
        if self.buf is None or self.buf.numel() < self.need:
            self._retired.append(self.buf)     # a captured step may hold the old one by address: never freed
            self.buf = torch.empty((self.need,), dtype=torch.uint8, device=self.device)
        return self.buf


class ParamArena:
    def __init__(self, device):
        self.device = device
        self.entries = []   # (name, shape, offset, numel)
        self.size = 0
        self.w = self.g = self.m = self.wb = None

    def register(self, name, shape):
        n = 1
        for s in shape:
            n *= s
        off = self.size
        self.entries.append((name, tuple(shape), off, n))
        self.size += (n + 63) // 64 * 64   # keep every tensor 256-B aligned in the fp32 arena
        return len(self.entries) - 1

    def finalize(self):
        dev = self.device
        self.w = torch.zeros((self.size,), dtype=torch.float32, device=dev)
        self.g = torch.zeros((self.size,), dtype=torch.float32, device=dev)
        self.m = torch.zeros((self.size,), dtype=torch.float32, device=dev)
        self.wb = torch.zeros((self.size,), dtype=torch.bfloat16, device=dev)

    def view(self, idx, which):
        _, shape, off, n = self.entries[idx]
        return getattr(self, which)[off:off + n].view(shape)

    def offset_of(self, idx):
        return self.entries[idx][2]

    def refresh_bf16(self):
        dense.f32_to_bf16(self.w, self.wb)

    def sgd_step(self, lr, momentum, wd, rescale):
        dense.sgd_momentum_update(self.w, self.g, self.m, self.wb, lr, momentum, wd, rescale)


class ConvLayer:
    """conv / fc with fused bias (+residual)(+ReLU); trainable or frozen.

    Filters are [Cout,KH,KW,Cin] bf16; a fully connected layer is KH=KW=1 on [R,1,1,Cin] tensors.
    """

    def __init__(self, name, cin, cout, k, stride=1, pad=None, bias=True, trainable=True, arena=None, ws=None,
                 device="cuda", gen=None, init_std=None, zero_init=False, train_bias=True, cout_real=None):
        self.name, self.cin, self.cout, self.k, self.stride = name, cin, cout, k, stride
        self.pad = (k // 2) if pad is None else pad
        self.trainable = trainable
        self.arena, self.ws = arena, ws
        self.has_bias = bias
        std = init_std if init_std is not None else math.sqrt(2.0 / (k * k * cin))   # He-normal
        w0 = torch.zeros((cout, k, k, cin)) if zero_init else torch.randn((cout, k, k, cin), generator=gen) * std
        # output channels past cout_real are alignment padding (fused / padded head outputs): zero filters and zero
        # biases, and every loss kernel writes zero gradients there, so they stay exactly zero through training
        self.cout_real = cout if cout_real is None else cout_real
        if self.cout_real < cout:
            w0[self.cout_real:] = 0
        self._w0 = w0
        self.train_bias = bias and trainable and train_bias
        self.frozen_bias = None
        if bias and not self.train_bias:   # folded frozen-BN shift: a constant
            self.frozen_bias = torch.zeros((cout,), dtype=torch.float32, device=device)
        if trainable:
            self.wi = arena.register(name + ".weight", (cout, k, k, cin))
            self.bi = arena.register(name + ".bias", (cout,)) if self.train_bias else None
        else:
            self.w_bf16 = w0.to(torch.bfloat16).to(device)
            self.bias_f32 = self.frozen_bias
        self.wt = None
        self.device = device
        # filters to warm while this layer's forward / data gradient runs (the layer executed next); see
        # mxdet_conv_desc_t.prefetch and ResNet.wire_prefetch
        self.pf_fwd = self.pf_bwd = None

    # called after arena.finalize()
    def materialize(self):
        if not self.trainable:
            return
        self.arena.view(self.wi, "w").copy_(self._w0.to(self.device))
        self.w_bf16 = self.arena.view(self.wi, "wb")
        self.bias_f32 = self.arena.view(self.bi, "w") if self.train_bias else self.frozen_bias
        self.wt = torch.empty((self.cin, self.k, self.k, self.cout), dtype=torch.bfloat16, device=self.device)
        self._w0 = None

    def plan(self, x_shape):
        """Reserve wgrad scratch for an input of this shape."""
        if self.trainable:
            self.ws.require(dense.conv2d_wgrad_workspace_bytes(x_shape, self.cout, self.k, self.k, self.stride, self.pad))

    def out_shape(self, x_shape):
        N, H, W, _ = x_shape
        return (N, (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1,
                self.cout)

    def refresh_transposed(self):
        if self.trainable:
            dense.filter_transpose(self.w_bf16, self.wt)

    def forward(self, x, relu=False, residual=None, res_upsample=False, out=None, bits_out=None):
        """bits_out: uint8 [N,Ho,Wo,Cout/8] receiving the 1-bit ReLU mask of the output (for the backward pass)."""
        if dense.PF_TRACE is not None:
            dense.PF_TRACE.append((self, "f", dense.mem_range(self.w_bf16), None))
        return dense.conv2d_forward(x, self.w_bf16, self.bias_f32, residual, self.stride, self.pad, relu, res_upsample,
                                    out, prefetch=self.pf_fwd, bits_out=bits_out)

    def fwd_call(self, x, relu=False, residual=None, res_upsample=False, out=None, bits_out=None):
        """Argument tuple of this layer's forward for dense.conv2d_group("fwd", ...)."""
        t = (x, self.w_bf16, self.bias_f32, residual, self.stride, self.pad, relu, res_upsample, out)
        return t if bits_out is None else t + (bits_out,)

    def dgrad_call(self, dy, x_shape, residual=None, relu_mask=None, accumulate=False, out=None, relu_bits=None):
        """Argument tuple of this layer's data gradient for dense.conv2d_group("dgrad", ...)."""
        t = (dy, self.wt, tuple(x_shape), self.k, self.k, self.stride, self.pad, residual, relu_mask, accumulate, out)
        return t if relu_bits is None else t + (relu_bits,)

    def backward_data(self, dy, x_shape, residual=None, relu_mask=None, accumulate=False, out=None, relu_bits=None):
        """relu_bits: the 1-bit form of relu_mask (read instead of it: 1/16 of the operand bytes)."""
        if dense.PF_TRACE is not None:
            dense.PF_TRACE.append((self, "b", dense.mem_range(self.wt), None))
        return dense.conv2d_dgrad(dy, self.wt, x_shape, self.k, self.k, self.stride, self.pad, residual,
                                  None if relu_bits is not None else relu_mask, accumulate, out, prefetch=self.pf_bwd,
                                  relu_bits=relu_bits)

    def backward_weight(self, x, dy, accumulate=False):
        if self.ws.grouping and not accumulate:
            self.ws.defer(self, x, dy)
            return
        dw = self.arena.view(self.wi, "g")
        db = self.arena.view(self.bi, "g") if self.train_bias else None
        ctx = self.ws.fork()
        if ctx is None:
            dense.conv2d_wgrad(x, dy, self.k, self.k, self.stride, self.pad, dw, db, accumulate, self.ws.get())
        else:
            with ctx:
                dense.conv2d_wgrad(x, dy, self.k, self.k, self.stride, self.pad, dw, db, accumulate, self.ws.get())
