"""Dense operators (MXNet roles: Convolution, FullyConnected, Pooling, UpSampling, sgd_mom_update) over the
HIP C-ABI. Tensors are torch CUDA tensors used purely as device memory: bf16 channels-last [N,H,W,C]
activations, bf16 [Cout,KH,KW,Cin] filters, fp32 gradients."""
import ctypes as C

import torch

from .. import _lib
from .._lib import ConvDescT, check, ptr, stream_ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


def mem_range(*tensors):
    """(ptr, bytes) of one tensor, or of the span of several when they are views of ONE allocation (the filters of a
    bottleneck's conv1 and projection shortcut are neighbours in the bf16 arena); else of the first. The kernels issue
    real loads over the span (mxdet_conv_desc_t.prefetch): a gap between two separate allocations is not the caller's
    memory and may not even be mapped, so separately allocated tensors are never spanned."""
    spans = [(t.data_ptr(), t.numel() * t.element_size()) for t in tensors]
    bases = {t.untyped_storage().data_ptr() for t in tensors}
    if len(bases) == 1:
        lo = min(p for p, _ in spans)
        hi = max(p + n for p, n in spans)
        if hi - lo <= 2 * sum(n for _, n in set(spans)):
            return (lo, hi - lo)
    return spans[0]


def conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, relu=False, res_upsample=False, accumulate=False, prefetch=None,
              relu_bits=None):
    d = ConvDescT()
    if relu_bits is not None:     # 1-bit ReLU mask (mxdet_conv_desc_t.relu_bits): uint8 [N,H,W,C/8]
        d.relu_bits = relu_bits.data_ptr()
    if prefetch is not None:      # the filter(s) of the launch that runs next (mxdet_conv_desc_t.prefetch): tensor or (ptr, bytes)
        d.prefetch, d.prefetch_bytes = prefetch if isinstance(prefetch, tuple) else mem_range(prefetch)
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = N, H, W, Cin, Cout, KH, KW, stride, pad
    d.Ho = (H + 2 * pad - KH) // stride + 1
    d.Wo = (W + 2 * pad - KW) // stride + 1
    d.relu, d.res_upsample, d.accumulate = int(relu), int(res_upsample), int(accumulate)
    return d


def conv2d_forward(x, w, bias=None, residual=None, stride=1, pad=0, relu=False, res_upsample=False, out=None, prefetch=None,
                   bits_out=None):
    """bits_out: uint8 [N,Ho,Wo,Cout/8] that receives the 1-bit (value > 0) mask of the output."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    d = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, relu, res_upsample, prefetch=prefetch, relu_bits=bits_out)
    if out is None:
        out = torch.empty((N, d.Ho, d.Wo, Cout), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_conv2d_fwd(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(out), stream_ptr()),
          "conv2d_fwd")
    return out


def conv2d_forward_splitk(x, w, bias=None, residual=None, relu=False, ksplit=4, out=None, workspace=None):
    """1x1 / stride-1 forward with the reduction split `ksplit` ways (long reductions on few rows: FC6)."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    d = conv_desc(N, H, W, Cin, Cout, 1, 1, 1, 0, relu, False)
    if out is None:
        out = torch.empty((N, d.Ho, d.Wo, Cout), dtype=torch.bfloat16, device=x.device)
    need = lib.mxdet_conv2d_fwd_splitk_workspace_bytes(C.byref(d), ksplit)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=x.device)
    check(lib.mxdet_conv2d_fwd_splitk(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(out), ksplit, ptr(workspace),
                                      workspace.numel(), stream_ptr()), "conv2d_fwd_splitk")
    return out


def conv2d_forward_chain(x, w, bias, w2, bias2=None, residual2=None, relu=True, relu2=True, out=None, prefetch=None,
                         w3=None, bias3=None, relu3=True, out3=None):
    """3x3 (stride 1, pad 1, Cout = 64) + 1x1 (64 -> 256) in one launch: relu2?(relu?(conv(x, w) + bias) * w2 + bias2 +
    residual2). The 64-channel intermediate map is never written (mxdet_conv2d_fwd_chain). w3 ([64,1,1,256]): a third
    convolution in the same launch, out3 = relu3?(out * w3 + bias3) (the next block's conv1); returns (out, out3) then."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cmid, Cout2 = w.shape[0], w2.shape[0]
    d = conv_desc(N, H, W, Cin, Cmid, 3, 3, 1, 1, relu, False, prefetch=prefetch)
    if out is None:
        out = torch.empty((N, H, W, Cout2), dtype=torch.bfloat16, device=x.device)
    if w3 is not None and out3 is None:
        out3 = torch.empty((N, H, W, w3.shape[0]), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_conv2d_fwd_chain(C.byref(d), ptr(x), ptr(w), ptr(bias), ptr(w2), ptr(bias2), Cout2, int(relu2),
                                     ptr(residual2), ptr(out), ptr(w3), ptr(bias3), 0 if w3 is None else w3.shape[0],
                                     int(relu3), ptr(out3), stream_ptr()), "conv2d_fwd_chain")
    return out if w3 is None else (out, out3)


def conv2d_dgrad(dy, wt, x_shape, KH, KW, stride=1, pad=0, residual=None, relu_mask=None, accumulate=False, out=None,
                 prefetch=None, relu_bits=None):
    """wt: [Cin,KH,KW,Cout] (filter_transpose of the forward filter). relu_bits: the 1-bit form of relu_mask
    (uint8 [N,H,W,Cin/8], written by conv2d_forward(bits_out=...)); if given it is read instead of relu_mask."""
    lib = _lib.load()
    N, H, W, Cin = x_shape
    Cout = dy.shape[3]
    d = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, relu=(relu_mask is not None or relu_bits is not None),
                  accumulate=accumulate, prefetch=prefetch, relu_bits=relu_bits)
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.bfloat16, device=dy.device)
    check(lib.mxdet_conv2d_dgrad(C.byref(d), ptr(dy), ptr(wt), ptr(residual), ptr(relu_mask), ptr(out), stream_ptr()),
          "conv2d_dgrad")
    return out


def conv2d_wgrad_workspace_bytes(x_shape, Cout, KH, KW, stride, pad):
    N, H, W, Cin = x_shape
    d = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad)
    return _lib.load().mxdet_conv2d_wgrad_workspace_bytes(C.byref(d))


def conv2d_wgrad(x, dy, KH, KW, stride=1, pad=0, dw=None, db=None, accumulate=False, workspace=None):
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    d = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, accumulate=accumulate)
    need = lib.mxdet_conv2d_wgrad_workspace_bytes(C.byref(d))
    if workspace is None:
        workspace = torch.empty((need,), dtype=torch.uint8, device=x.device)
    if dw is None:
        dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.float32, device=x.device)
    check(lib.mxdet_conv2d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(workspace), workspace.numel(),
                                 stream_ptr()), "conv2d_wgrad")
    return dw


class GroupedConv:
    """Independent convolutions of one kind as ONE launch (mxdet_conv2d_grouped). kind "fwd": calls are the argument
    tuples of conv2d_forward (x, w, bias, residual, stride, pad, relu, res_upsample, out); kind "dgrad": those of
    conv2d_dgrad (dy, wt, x_shape, KH, KW, stride, pad, residual, relu_mask, accumulate, out); `out` is required. The
    plan holds device addresses: build it once in eager mode, relaunch while the tensors stay where they are."""

    def __init__(self, kind, calls, device, hint=None):
        """hint: (ptr, bytes) the launch after this one reads first; the group's first item carries it."""
        lib = _lib.load()
        n = len(calls)
        items = (_lib.ConvItemT * n)()
        self.keep, self.kind = calls, 0 if kind == "fwd" else 1
        self.flops = 0.0
        dp = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        for it, c in zip(items, calls):
            if self.kind == 0:
                x, w, bias, residual, stride, pad, relu, res_up, out = c[:9]
                bits = c[9] if len(c) > 9 else None
                N, H, W, Cin = x.shape
                Cout, KH, KW, _ = w.shape
                it.desc = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, relu, res_up, prefetch=hint, relu_bits=bits)
                hint = None
                it.src, it.filt, it.bias, it.residual, it.relu_mask, it.dst = dp(x), dp(w), dp(bias), dp(residual), None, dp(out)
                self.flops += 2.0 * N * it.desc.Ho * it.desc.Wo * Cout * KH * KW * Cin
            else:
                dy, wt, x_shape, KH, KW, stride, pad, residual, relu_mask, accumulate, out = c[:11]
                bits = c[11] if len(c) > 11 else None            # 1-bit form of relu_mask, read instead of it
                N, H, W, Cin = x_shape
                Cout = dy.shape[3]
                it.desc = conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad, relu=(relu_mask is not None or bits is not None),
                                    accumulate=accumulate, prefetch=hint, relu_bits=bits)
                hint = None
                it.src, it.filt, it.bias, it.residual, it.relu_mask, it.dst = (dp(dy), dp(wt), None, dp(residual),
                                                                              None if bits is not None else dp(relu_mask), dp(out))
                self.flops += 2.0 * N * dy.shape[1] * dy.shape[2] * Cout * KH * KW * Cin
        nbytes = lib.mxdet_conv2d_grouped_table_bytes(n)
        host = (C.c_ubyte * nbytes)()
        cfg, grid = C.c_int32(0), C.c_int32(0)
        check(lib.mxdet_conv2d_grouped_plan(items, n, self.kind, host, nbytes, C.byref(cfg), C.byref(grid)),
              "conv2d_grouped_plan")
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).clone().to(device)
        self.n, self.cfg, self.grid = n, cfg.value, grid.value

    def launch(self):
        check(_lib.load().mxdet_conv2d_grouped(ptr(self.table), self.n, self.kind, self.cfg, self.grid, stream_ptr()),
              "conv2d_grouped")


_group_plans = {}


# Issue-order trace of one eager step, for the prefetch hints (mxdet_conv_desc_t.prefetch): entries are
# (layer or None, "f" | "b" | "g", (ptr, bytes) of the filter(s) the launch reads, plan key of a grouped launch).
# models/utils/detector.py wires every launch's hint to the filters of the launch that follows it; a grouped launch's
# hint lives in GROUP_HINTS (its plan is rebuilt with it at the next eager call).
PF_TRACE = None
GROUP_HINTS = {}


def conv2d_group(kind, calls, device):
    """Run `calls` (see GroupedConv) as one grouped launch; plans are cached by the tensors' addresses. Under stream
    capture an unseen group cannot upload its table: it falls back to one launch per call (same results)."""
    key = (kind,) + tuple(tuple(t.data_ptr() if torch.is_tensor(t) else t for t in c) for c in calls)
    if PF_TRACE is not None and calls:
        PF_TRACE.append((None, "g", mem_range(*[c[1] for c in calls]), key if len(calls) > 1 else None))
    if len(calls) == 1:
        plan = None
    else:
        plan = _group_plans.get(key)
        if plan is None and not torch.cuda.is_current_stream_capturing():
            plan = GroupedConv(kind, calls, device, GROUP_HINTS.get(key))
            _group_plans[key] = plan
    if plan is not None:
        plan.launch()
        return
    for c in calls:
        if kind == "fwd":
            x, w, bias, residual, stride, pad, relu, res_up, out = c[:9]
            conv2d_forward(x, w, bias, residual, stride, pad, relu, res_up, out, bits_out=c[9] if len(c) > 9 else None)
        else:
            dy, wt, x_shape, KH, KW, stride, pad, residual, relu_mask, accumulate, out = c[:11]
            conv2d_dgrad(dy, wt, x_shape, KH, KW, stride, pad, residual, relu_mask, accumulate, out,
                         relu_bits=c[11] if len(c) > 11 else None)


class GroupedWgrad:
    """Weight gradients of several layers as one launch pair (mxdet_conv2d_wgrad_grouped). `calls` is a list of
    (x, dy, KH, KW, stride, pad, dw, db, accumulate); the plan (device table, grids, workspace size) is built once, in
    eager mode, and stays valid while the tensors keep their addresses (e.g. under hipGraph replay)."""

    def __init__(self, calls, device):
        lib = _lib.load()
        n = len(calls)
        items = (_lib.WgradItemT * n)()
        self.keep = calls                     # the tensors whose addresses the table holds
        self.flops = 0.0
        self.heaviest = (0.0, "")
        for it, (x, dy, KH, KW, stride, pad, dw, db, accumulate) in zip(items, calls):
            N, H, W, Cin = x.shape
            it.desc = conv_desc(N, H, W, Cin, dy.shape[3], KH, KW, stride, pad, accumulate=accumulate)
            it.x, it.dy, it.dw, it.db = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
            fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * KH * KW * Cin
            self.flops += fl
            if fl > self.heaviest[0]:
                self.heaviest = (fl, "N=%d %dx%d %d->%d %dx%d" % (N, H, W, Cin, dy.shape[3], KH, KW))
        nbytes = lib.mxdet_conv2d_wgrad_grouped_table_bytes(n)
        host = (C.c_ubyte * nbytes)()
        ws, gw, gb, gr = C.c_size_t(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib.mxdet_conv2d_wgrad_grouped_plan(items, n, host, nbytes, C.byref(ws), C.byref(gw), C.byref(gb),
                                                  C.byref(gr)), "conv2d_wgrad_grouped_plan")
        self.grid_big = gb.value
        self.table = torch.frombuffer(bytearray(host), dtype=torch.uint8).clone().to(device)
        self.n, self.grid_wgrad, self.grid_reduce, self.workspace_bytes = n, gw.value, gr.value, ws.value

    def launch(self, workspace):
        check(_lib.load().mxdet_conv2d_wgrad_grouped(ptr(self.table), self.n, self.grid_wgrad, self.grid_big, self.grid_reduce,
                                                     ptr(workspace), workspace.numel() if workspace is not None else 0,
                                                     self.workspace_bytes, stream_ptr()), "conv2d_wgrad_grouped")


def filter_transpose(w, out=None):
    lib = _lib.load()
    Cout, KH, KW, Cin = w.shape
    if out is None:
        out = torch.empty((Cin, KH, KW, Cout), dtype=torch.bfloat16, device=w.device)
    check(lib.mxdet_filter_transpose(ptr(w), Cout, KH, KW, Cin, ptr(out), stream_ptr()), "filter_transpose")
    return out


def make_transpose_table(pairs, device):
    """pairs: list of (w [Cout,KH,KW,Cin] bf16, wt [Cin,KH,KW,Cout] bf16). Returns (device table, ndesc, total_tiles)."""
    import struct
    buf = bytearray()
    tile0 = 0
    for w, wt in pairs:
        Cout, KH, KW, Cin = w.shape
        buf += struct.pack("<QQiiii", w.data_ptr(), wt.data_ptr(), Cout, KH * KW, Cin, tile0)
        tile0 += ((Cin + 63) // 64) * ((Cout + 63) // 64) * KH * KW
    table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(device)
    return table, len(pairs), tile0


def filter_transpose_batched(table, ndesc, total_tiles):
    check(_lib.load().mxdet_filter_transpose_batched(ptr(table), ndesc, total_tiles, stream_ptr()),
          "filter_transpose_batched")


def stem_conv7x7(image, w, bias=None, out=None):
    """image: NCHW [N,3,H,W] f32|bf16; w bf16 [64,7,7,3]; returns bf16 [N,Ho,Wo,64] (conv + bias + ReLU)."""
    lib = _lib.load()
    N, _, H, W = image.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=image.device)
    check(lib.mxdet_stem_conv7x7(ptr(image), _DT[image.dtype], N, H, W, ptr(w), ptr(bias), ptr(out), stream_ptr()),
          "stem_conv7x7")
    return out


def stem_conv7x7_pool(image, w, bias=None, out=None):
    """Stem convolution + bias + ReLU + 3x3/2 max-pool in one pass: NCHW image -> bf16 [N,Hp,Wp,64]."""
    lib = _lib.load()
    N, _, H, W = image.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Hp, Wp, 64), dtype=torch.bfloat16, device=image.device)
    check(lib.mxdet_stem_conv7x7_pool(ptr(image), _DT[image.dtype], N, H, W, ptr(w), ptr(bias), ptr(out), stream_ptr()),
          "stem_conv7x7_pool")
    return out


def maxpool3x3s2(x, out=None):
    lib = _lib.load()
    N, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, Cc), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_maxpool3x3s2(ptr(x), N, H, W, Cc, ptr(out), stream_ptr()), "maxpool3x3s2")
    return out


def subsample2(x, out=None):
    lib = _lib.load()
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, Cc), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_subsample2(ptr(x), N, H, W, Cc, ptr(out), stream_ptr()), "subsample2")
    return out


def subsample2_backward(dy, dx, accumulate=False):
    lib = _lib.load()
    N, H, W, Cc = dx.shape
    check(lib.mxdet_subsample2_bwd(ptr(dy), N, H, W, Cc, int(accumulate), ptr(dx), stream_ptr()), "subsample2_bwd")
    return dx


def upsample2_backward(dfine, dcoarse, accumulate=False):
    lib = _lib.load()
    N, Hf, Wf, Cc = dfine.shape
    check(lib.mxdet_upsample2_bwd(ptr(dfine), N, Hf, Wf, Cc, int(accumulate), ptr(dcoarse), stream_ptr()),
          "upsample2_bwd")
    return dcoarse


def pixel_shuffle2_inv_relu(dy_up, act, out=None):
    """Backward of conv1x1 -> ReLU -> pixel shuffle: dy_up [R,2H,2W,C], act (packed activation) [R,H,W,4C] -> [R,H,W,4C]."""
    R, H, W, C4 = act.shape
    if out is None:
        out = torch.empty_like(act)
    check(_lib.load().mxdet_pixel_shuffle2_inv_relu(ptr(dy_up), ptr(act), R, H, W, C4 // 4, ptr(out), stream_ptr()),
          "pixel_shuffle2_inv_relu")
    return out


def pixel_shuffle2(x, out=None, inverse=False):
    """[R,H,W,4C] -> [R,2H,2W,C] (or back with inverse=True): the data movement of a 2x2 stride-2 deconvolution."""
    lib = _lib.load()
    if not inverse:
        R, H, W, C4 = x.shape
        Cc = C4 // 4
        if out is None:
            out = torch.empty((R, 2 * H, 2 * W, Cc), dtype=torch.bfloat16, device=x.device)
    else:
        R, H2, W2, Cc = x.shape
        H, W = H2 // 2, W2 // 2
        if out is None:
            out = torch.empty((R, H, W, 4 * Cc), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_pixel_shuffle2(ptr(x), R, H, W, Cc, int(inverse), ptr(out), stream_ptr()), "pixel_shuffle2")
    return out


def add_bf16(a, b, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(a)
    check(lib.mxdet_add_bf16(ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr()), "add_bf16")
    return out


def relu_backward(dy, y, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(dy)
    check(lib.mxdet_relu_bwd_bf16(ptr(dy), ptr(y), dy.numel(), ptr(out), stream_ptr()), "relu_bwd_bf16")
    return out


def relu_forward(x, out=None):
    """y = max(x, 0) (bf16): the ReLU-backward kernel with dy = y = x passes x where x > 0."""
    return relu_backward(x, x, out)


def f32_to_bf16(x, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_f32_to_bf16(ptr(x), x.numel(), ptr(out), stream_ptr()), "f32_to_bf16")
    return out


def f32_accum_to_bf16(x, y, accumulate):
    check(_lib.load().mxdet_f32_accum_to_bf16(ptr(x), x.numel(), int(accumulate), ptr(y), stream_ptr()),
          "f32_accum_to_bf16")
    return y


def nchw_to_nhwc(x):
    lib = _lib.load()
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device=x.device)
    check(lib.mxdet_nchw_to_nhwc_bf16(ptr(x), _DT[x.dtype], N, Cc, H, W, ptr(out), stream_ptr()), "nchw_to_nhwc_bf16")
    return out


def nhwc_to_nchw(x):
    lib = _lib.load()
    N, H, W, Cc = x.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    check(lib.mxdet_nhwc_to_nchw_f32(ptr(x), N, Cc, H, W, ptr(out), stream_ptr()), "nhwc_to_nchw_f32")
    return out


def sgd_momentum_update(w, grad, mom, w_bf16, lr, momentum=0.9, wd=1e-4, rescale=1.0):
    """lr: python float, or a 1-element fp32 device tensor read when the kernel runs (scheduled lr under hipGraph replay)."""
    if torch.is_tensor(lr):
        check(_lib.load().mxdet_sgd_momentum_update_sched(ptr(w), ptr(grad), ptr(mom), ptr(w_bf16), w.numel(), ptr(lr),
                                                          momentum, wd, rescale, stream_ptr()), "sgd_momentum_update_sched")
        return
    check(_lib.load().mxdet_sgd_momentum_update(ptr(w), ptr(grad), ptr(mom), ptr(w_bf16), w.numel(), lr, momentum, wd,
                                                rescale, stream_ptr()), "sgd_momentum_update")
