"""numpy front-end of the C oracle (oracle/mxdet_oracle.c). TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED by the reference: /root/reference holds no code or fixtures (see mxdet_oracle.c header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmxdet_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.oracle_expf.restype = C.c_float
        _lib.oracle_expf.argtypes = [C.c_float]
        _lib.oracle_logf.restype = C.c_float
        _lib.oracle_logf.argtypes = [C.c_float]
        _lib.oracle_sample_key.restype = C.c_uint32
        _lib.oracle_sample_key.argtypes = [C.c_uint32] * 5
        _lib.oracle_f32_to_bf16.restype = C.c_uint16
        _lib.oracle_f32_to_bf16.argtypes = [C.c_float]
        _lib.oracle_fpn_level.restype = C.c_int
        _lib.oracle_nms.restype = C.c_int
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- bf16 helpers (numpy has no bf16: keep the bits in uint16) -----------------------------------
def f32_to_bf16_bits(x):
    x = _c(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b):
    return (_c(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


def round_bf16(x):
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# ---- scalar helpers -------------------------------------------------------------------------------
def expf(x):
    L = lib()
    return np.array([L.oracle_expf(float(v)) for v in np.ravel(x)], dtype=np.float32).reshape(np.shape(x))


def logf(x):
    L = lib()
    return np.array([L.oracle_logf(float(v)) for v in np.ravel(x)], dtype=np.float32).reshape(np.shape(x))


def sample_key(seed, step, image, stream, idx):
    return lib().oracle_sample_key(seed, step, image, stream, idx)


def philox(c, k):
    out = np.zeros(4, np.uint32)
    lib().oracle_philox(C.c_uint32(c[0]), C.c_uint32(c[1]), C.c_uint32(c[2]), C.c_uint32(c[3]), C.c_uint32(k[0]),
                        C.c_uint32(k[1]), _vp(out))
    return out


def decode_clip(box, d, im_h, im_w):
    out = np.zeros(4, np.float32)
    lib().oracle_decode_clip(_vp(_c(box, np.float32)), _vp(_c(d, np.float32)), C.c_float(im_h), C.c_float(im_w),
                             _vp(out))
    return out


def encode(ex, gt):
    out = np.zeros(4, np.float32)
    lib().oracle_encode(_vp(_c(ex, np.float32)), _vp(_c(gt, np.float32)), _vp(out))
    return out


# ---- boxes / anchors ------------------------------------------------------------------------------
def box_iou(a, b):
    a, b = _c(a, np.float32).reshape(-1, 4), _c(b, np.float32).reshape(-1, 4)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_box_iou(_vp(a), C.c_int64(a.shape[0]), _vp(b), C.c_int64(b.shape[0]), _vp(out))
    return out


def base_anchors(stride, ratios=(0.5, 1.0, 2.0), scales=(8.0,)):
    r, s = _c(ratios, np.float64), _c(scales, np.float64)
    out = np.zeros((len(r) * len(s), 4), np.float32)
    lib().oracle_base_anchors(C.c_int(stride), _vp(r), C.c_int(len(r)), _vp(s), C.c_int(len(s)), _vp(out))
    return out


def grid_anchors(base, H, W, stride):
    base = _c(base, np.float32)
    A = base.shape[0]
    out = np.zeros((H * W * A, 4), np.float32)
    lib().oracle_grid_anchors(_vp(base), C.c_int(A), C.c_int(H), C.c_int(W), C.c_int(stride), _vp(out))
    return out


def fpn_level(rois, lvl_min=2, lvl_max=5):
    rois = _c(rois, np.float32).reshape(-1, 5)
    L = lib()
    return np.array([L.oracle_fpn_level(_vp(rois[i]), C.c_int(lvl_min), C.c_int(lvl_max)) for i in range(len(rois))],
                    dtype=np.int32)


def nms(boxes, thresh, max_keep=None, invalid=None):
    boxes = _c(boxes, np.float32).reshape(-1, 4)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    inv = None if invalid is None else _c(invalid, np.uint8)
    k = lib().oracle_nms(_vp(boxes), C.c_int(n), _vp(inv), C.c_float(thresh), C.c_int(n if max_keep is None else max_keep),
                         _vp(keep))
    return keep[:k].copy()


def proposal(scores, deltas, base, H, W, strides, im_info, pre_n, post_n, thresh, min_size):
    """scores[l]: [N, H*W*A] f32; deltas[l]: [N, H*W*A, 4] f32 (canonical (y,x,a) order)."""
    L = len(scores)
    A = base[0].shape[0]
    N = scores[0].shape[0]
    sc = [_c(s, np.float32) for s in scores]
    dl = [_c(d, np.float32) for d in deltas]
    bs = [_c(b, np.float32) for b in base]
    Pf = C.c_void_p * L
    rois = np.zeros((N, post_n, 5), np.float32)
    rs = np.zeros((N, post_n), np.float32)
    ra = np.zeros((N, post_n), np.int32)
    nr = np.zeros((N,), np.int32)
    lib().oracle_proposal(C.c_int(L), C.c_int(A), _vp(_c(H, np.int32)), _vp(_c(W, np.int32)), _vp(_c(strides, np.int32)),
                          Pf(*[s.ctypes.data for s in sc]), Pf(*[d.ctypes.data for d in dl]),
                          Pf(*[b.ctypes.data for b in bs]), C.c_int(N), _vp(_c(im_info, np.float32)), C.c_int(pre_n),
                          C.c_int(post_n), C.c_float(thresh), C.c_float(min_size), _vp(rois), _vp(rs), _vp(ra), _vp(nr))
    return rois, rs, ra, nr


def anchor_target(anchors, gt, im_info, fg_thresh=0.7, bg_thresh=0.3, border=0.0, batch_size=256, fg_fraction=0.5,
                  seed=0, step=0, image_offset=0):
    anchors = _c(anchors, np.float32)
    gt = _c(gt, np.float32)
    N, G = gt.shape[0], gt.shape[1]
    A = anchors.shape[0]
    labels = np.zeros((N, A), np.int32)
    matched = np.zeros((N, A), np.int32)
    targets = np.zeros((N, A, 4), np.float32)
    miou = np.zeros((N, A), np.float32)
    lib().oracle_anchor_target(_vp(anchors), C.c_int64(A), _vp(gt), C.c_int(N), C.c_int(G),
                               _vp(_c(im_info, np.float32)), C.c_float(fg_thresh), C.c_float(bg_thresh),
                               C.c_float(border), C.c_int(batch_size), C.c_float(fg_fraction), C.c_uint32(seed),
                               C.c_uint32(step), C.c_uint32(image_offset), _vp(labels), _vp(matched), _vp(targets),
                               _vp(miou))
    return labels, matched, targets, miou


def proposal_target(rois, num_rois, gt, R=512, fg_fraction=0.25, fg_thresh=0.5, bg_hi=0.5, bg_lo=0.0, num_classes=81,
                    class_agnostic=False, means=(0, 0, 0, 0), stds=(0.1, 0.1, 0.2, 0.2), seed=0, step=0,
                    image_offset=0):
    rois = _c(rois, np.float32)
    gt = _c(gt, np.float32)
    N, S = rois.shape[0], rois.shape[1]
    G = gt.shape[1]
    D = 4 if class_agnostic else 4 * num_classes
    out_rois = np.zeros((N, R, 5), np.float32)
    labels = np.zeros((N, R), np.int32)
    tgt = np.zeros((N, R, D), np.float32)
    wgt = np.zeros((N, R, D), np.float32)
    matched = np.zeros((N, R), np.int32)
    nfg = np.zeros((N,), np.int32)
    lib().oracle_proposal_target(_vp(rois), _vp(_c(num_rois, np.int32)), C.c_int(S), _vp(gt), C.c_int(N), C.c_int(G),
                                 C.c_int(R), C.c_float(fg_fraction), C.c_float(fg_thresh), C.c_float(bg_hi),
                                 C.c_float(bg_lo), C.c_int(num_classes), C.c_int(int(class_agnostic)),
                                 _vp(_c(means, np.float32)), _vp(_c(stds, np.float32)), C.c_uint32(seed),
                                 C.c_uint32(step), C.c_uint32(image_offset), _vp(out_rois), _vp(labels), _vp(tgt),
                                 _vp(wgt), _vp(matched), _vp(nfg))
    return out_rois, labels, tgt, wgt, matched, nfg


def roi_align(feats_bits, scales, rois, levels, PH=7, PW=7, sampling_ratio=2, lvl_min=2, grad_out_bits=None):
    """feats_bits[l]: uint16 bf16 bits [N,H,W,C]. Forward returns bf16 bits [R,PH,PW,C]; with grad_out_bits
    returns the list of fp32 gradient maps."""
    L = len(feats_bits)
    fb = [_c(f, np.uint16) for f in feats_bits]
    N, C_ = fb[0].shape[0], fb[0].shape[3]
    H = _c([f.shape[1] for f in fb], np.int32)
    W = _c([f.shape[2] for f in fb], np.int32)
    rois = _c(rois, np.float32).reshape(-1, 5)
    levels = _c(levels, np.int32)
    R = rois.shape[0]
    Pf = C.c_void_p * L
    if grad_out_bits is None:
        out = np.zeros((R, PH, PW, C_), np.uint16)
        lib().oracle_roi_align(C.c_int(L), C.c_int(lvl_min), _vp(H), _vp(W), _vp(_c(scales, np.float32)),
                               Pf(*[f.ctypes.data for f in fb]), None, C.c_int(N), C.c_int(C_), _vp(rois), _vp(levels),
                               C.c_int64(R), C.c_int(PH), C.c_int(PW), C.c_int(sampling_ratio), _vp(out), None,
                               C.c_int(0))
        return out
    go = _c(grad_out_bits, np.uint16)
    df = [np.zeros(f.shape, np.float32) for f in fb]
    lib().oracle_roi_align(C.c_int(L), C.c_int(lvl_min), _vp(H), _vp(W), _vp(_c(scales, np.float32)),
                           Pf(*[f.ctypes.data for f in fb]), Pf(*[d.ctypes.data for d in df]), C.c_int(N), C.c_int(C_),
                           _vp(rois), _vp(levels), C.c_int64(R), C.c_int(PH), C.c_int(PW), C.c_int(sampling_ratio),
                           None, _vp(go), C.c_int(1))
    return df


# ---- losses -----------------------------------------------------------------------------------------
def smooth_l1(p, t, w=None, sigma=1.0):
    p, t = _c(p, np.float32), _c(t, np.float32)
    out, grad = np.zeros_like(p), np.zeros_like(p)
    ww = None if w is None else _c(w, np.float32)
    lib().oracle_smooth_l1(_vp(p), _vp(t), _vp(ww), C.c_int64(p.size), C.c_float(sigma), _vp(out), _vp(grad))
    return out, grad


def rpn_loss_level(head, A, labels, targets, level_offset, sigma, norm, loss_scale):
    head = _c(head, np.float32)
    N, H, W, Cp = head.shape
    grad = np.zeros_like(head)
    loss = np.zeros(2, np.float64)
    labels = _c(labels, np.int32)
    lib().oracle_rpn_loss_level(_vp(head), C.c_int(N), C.c_int(H), C.c_int(W), C.c_int(A), C.c_int(Cp), _vp(labels),
                                _vp(_c(targets, np.float32)), C.c_int64(labels.shape[1]), C.c_int64(level_offset),
                                C.c_float(sigma), C.c_float(norm), C.c_float(loss_scale), _vp(grad), _vp(loss))
    return loss, grad


def rcnn_loss(cls, reg, labels, tgt, wgt, num_classes, reg_dim, sigma, norm, loss_scale):
    cls, reg = _c(cls, np.float32), _c(reg, np.float32)
    R = cls.shape[0]
    gc, gr = np.zeros_like(cls), np.zeros_like(reg)
    loss = np.zeros(2, np.float64)
    lib().oracle_rcnn_loss(_vp(cls), _vp(reg), C.c_int(cls.shape[1]), C.c_int(reg.shape[1]), _vp(_c(labels, np.int32)),
                           _vp(_c(tgt, np.float32)), _vp(_c(wgt, np.float32)), C.c_int64(R), C.c_int(num_classes),
                           C.c_int(reg_dim), C.c_float(sigma), C.c_float(norm), C.c_float(loss_scale), _vp(gc), _vp(gr),
                           _vp(loss))
    return loss, gc, gr


def focal_loss(logits, labels, alpha=0.25, gamma=2.0, grad_scale=1.0):
    logits = _c(logits, np.float32)
    n, Cc = logits.shape
    grad = np.zeros_like(logits)
    loss = np.zeros(1, np.float64)
    lib().oracle_focal_loss(_vp(logits), _vp(_c(labels, np.int32)), C.c_int64(n), C.c_int(Cc), C.c_float(alpha),
                            C.c_float(gamma), C.c_float(grad_scale), _vp(grad), _vp(loss))
    return loss, grad


def mask_target(rois, matched, labels, masks, S=28):
    rois = _c(rois, np.float32).reshape(-1, 5)
    masks = _c(masks, np.uint8)
    N, G, H, W = masks.shape
    R = rois.shape[0]
    tg = np.zeros((R, S, S), np.uint8)
    cls = np.zeros((R,), np.int32)
    lib().oracle_mask_target(_vp(rois), _vp(_c(matched, np.int32)), _vp(_c(labels, np.int32)), _vp(masks), C.c_int64(R),
                             C.c_int(G), C.c_int(H), C.c_int(W), C.c_int(S), _vp(tg), _vp(cls))
    return tg, cls


def mask_loss(logits, cls, targets, loss_scale=1.0):
    logits = _c(logits, np.float32)
    R, S, _, Cp = logits.shape
    grad = np.zeros_like(logits)
    loss = np.zeros(1, np.float64)
    lib().oracle_mask_loss(_vp(logits), _vp(_c(cls, np.int32)), _vp(_c(targets, np.uint8)), C.c_int64(R), C.c_int(S),
                           C.c_int(Cp), C.c_float(loss_scale), _vp(grad), _vp(loss))
    return loss, grad


# ---- dense ------------------------------------------------------------------------------------------
def conv2d_fwd(x, w, bias=None, residual=None, stride=1, pad=0, relu=False, res_upsample=False):
    x, w = _c(x, np.float32), _c(w, np.float32)
    N, H, W_, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    y = np.zeros((N, Ho, Wo, Cout), np.float32)
    b = None if bias is None else _c(bias, np.float32)
    r = None if residual is None else _c(residual, np.float32)
    lib().oracle_conv2d_fwd(_vp(x), _vp(w), _vp(b), _vp(r), C.c_int(N), C.c_int(H), C.c_int(W_), C.c_int(Cin),
                            C.c_int(Cout), C.c_int(KH), C.c_int(KW), C.c_int(stride), C.c_int(pad), C.c_int(Ho),
                            C.c_int(Wo), C.c_int(int(relu)), C.c_int(int(res_upsample)), _vp(y))
    return y


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0):
    dy, w = _c(dy, np.float32), _c(w, np.float32)
    N, H, W_, Cin = x_shape
    Cout, KH, KW, _ = w.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    dx = np.zeros(x_shape, np.float32)
    lib().oracle_conv2d_dgrad(_vp(dy), _vp(w), C.c_int(N), C.c_int(H), C.c_int(W_), C.c_int(Cin), C.c_int(Cout),
                              C.c_int(KH), C.c_int(KW), C.c_int(stride), C.c_int(pad), C.c_int(Ho), C.c_int(Wo), _vp(dx))
    return dx


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0):
    x, dy = _c(x, np.float32), _c(dy, np.float32)
    N, H, W_, Cin = x.shape
    Cout, KH, KW, _ = w_shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    dw = np.zeros(w_shape, np.float32)
    db = np.zeros((Cout,), np.float32)
    lib().oracle_conv2d_wgrad(_vp(x), _vp(dy), C.c_int(N), C.c_int(H), C.c_int(W_), C.c_int(Cin), C.c_int(Cout),
                              C.c_int(KH), C.c_int(KW), C.c_int(stride), C.c_int(pad), C.c_int(Ho), C.c_int(Wo), _vp(dw),
                              _vp(db))
    return dw, db


def maxpool3x3s2(x):
    x = _c(x, np.float32)
    N, H, W_, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W_ + 2 - 3) // 2 + 1
    y = np.zeros((N, Ho, Wo, Cc), np.float32)
    lib().oracle_maxpool3x3s2(_vp(x), C.c_int(N), C.c_int(H), C.c_int(W_), C.c_int(Cc), _vp(y))
    return y


def detection_postprocess(cls, reg, rois, num_rois, im_info, means, stds, score_thresh, nms_thresh, max_det):
    """cls [N*R, C] f32, reg [N*R, 4C] f32, rois [N*R, 5]. Returns dets [N,max_det,6], num [N], scores [N*R,C], boxes [N*R,C,4]."""
    cls, reg, rois = _c(cls, np.float32), _c(reg, np.float32), _c(rois, np.float32)
    N = int(np.asarray(im_info).shape[0])
    Cn = cls.shape[1]
    R = cls.shape[0] // N
    dets = np.zeros((N, max_det, 6), np.float32)
    num = np.zeros((N,), np.int32)
    sc = np.zeros((N * R, Cn), np.float32)
    bb = np.zeros((N * R, Cn, 4), np.float32)
    lib().oracle_detection_postprocess(_vp(cls), _vp(reg), _vp(rois), _vp(_c(num_rois, np.int32)), _vp(_c(im_info, np.float32)),
                                       C.c_int(N), C.c_int(R), C.c_int(Cn), _vp(_c(means, np.float32)), _vp(_c(stds, np.float32)),
                                       C.c_float(score_thresh), C.c_float(nms_thresh), C.c_int(max_det), _vp(dets), _vp(num),
                                       _vp(sc), _vp(bb))
    return dets, num, sc, bb


def image_preprocess(images, scales, flips, Hp, Wp, mean, std, swap_rb=True, return_u8=False):
    """images: list of u8 [h,w,3]; scales: list of float (python double). Returns bf16 bits [N,3,Hp,Wp] (+ the 8-bit
    resize results). dst size = rint(src * scale) (round half even, = cv2's saturate_cast<int>)."""
    N = len(images)
    out = np.zeros((N, 3, Hp, Wp), np.uint16)
    u8 = []
    for n, im in enumerate(images):
        im = _c(im, np.uint8)
        sh, sw = im.shape[:2]
        dh, dw = int(round(sh * scales[n])), int(round(sw * scales[n]))
        r8 = np.zeros((dh, dw, 3), np.uint8)
        lib().oracle_image_preprocess(_vp(im), C.c_int(sh), C.c_int(sw), C.c_int(dh), C.c_int(dw), C.c_int(int(flips[n])),
                                      C.c_double(1.0 / scales[n]), C.c_int(Hp), C.c_int(Wp), _vp(_c(mean, np.float32)),
                                      _vp(_c(std, np.float32)), C.c_int(int(swap_rb)), _vp(out[n]), _vp(r8))
        u8.append(r8)
    return (out, u8) if return_u8 else out


def polygon_masks(verts, poly_start, inst_first, NG, H, W):
    verts, poly_start, inst_first = _c(verts, np.float32), _c(poly_start, np.int32), _c(inst_first, np.int32)
    masks = np.zeros((NG, H, W), np.uint8)
    lib().oracle_polygon_masks(_vp(verts), _vp(poly_start), _vp(inst_first), C.c_int(NG), C.c_int(H), C.c_int(W), _vp(masks))
    return masks


def mask_paste(logits, dets, H, W, thresh=0.5):
    """logits [R,S,S,Cpad] f32 (bf16-valued), dets [R,6] -> masks [R,H,W] u8."""
    logits, dets = _c(logits, np.float32), _c(dets, np.float32)
    R, S, _, Cpad = logits.shape
    masks = np.zeros((R, H, W), np.uint8)
    lib().oracle_mask_paste(_vp(logits), _vp(dets), C.c_int(R), C.c_int(S), C.c_int(Cpad), C.c_int(H), C.c_int(W),
                            C.c_float(thresh), _vp(masks))
    return masks


def class_nms_topk(boxes, scores, cls, num, C_, score_thresh, nms_thresh, max_det):
    boxes, scores, cls = _c(boxes, np.float32), _c(scores, np.float32), _c(cls, np.int32)
    N, R = scores.shape
    dets = np.zeros((N, max_det, 6), np.float32)
    nd = np.zeros((N,), np.int32)
    lib().oracle_class_nms_topk(_vp(boxes), _vp(scores), _vp(cls), _vp(_c(num, np.int32)), C.c_int(N), C.c_int(R), C.c_int(C_),
                                C.c_float(score_thresh), C.c_float(nms_thresh), C.c_int(max_det), _vp(dets), _vp(nd))
    return dets, nd


def retina_detect(cls_logits, deltas, base, H, W, strides, im_info, num_classes, pre_n=1000, score_thresh=0.05, nms_thresh=0.5,
                  max_det=100, cap=4096):
    """cls_logits[l]: [N, H*W*A*C] f32 ((y,x,a,c) order); deltas[l]: [N, H*W*A, 4]; base[l]: [A,4]. Restates
    mxdet_retina_detect: per-level top-k over (anchor, class) -> decode -> merge (cut to `cap`) -> sigmoid -> per-class NMS."""
    Cn = num_classes
    exp_d = [np.repeat(_c(d, np.float32), Cn, axis=1) for d in deltas]
    exp_b = [np.repeat(_c(b, np.float32), Cn, axis=0) for b in base]
    L = len(cls_logits)
    post = min(L * pre_n, cap)
    rois, logit, gidx, num = proposal(cls_logits, exp_d, exp_b, H, W, strides, im_info, pre_n, post, 2.0, 0.0)
    lib().oracle_sigmoid.restype = C.c_float
    lib().oracle_sigmoid.argtypes = [C.c_float]
    prob = np.array([[lib().oracle_sigmoid(float(z)) for z in row] for row in logit], np.float32)
    cls = (gidx % Cn + 1).astype(np.int32)
    for n in range(rois.shape[0]):
        cls[n, num[n]:] = 0
    return class_nms_topk(rois[:, :, 1:5], prob, cls, num, Cn, score_thresh, nms_thresh, max_det)
