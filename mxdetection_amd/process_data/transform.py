"""Resize-to-800/1333, horizontal flip, mean/std normalisation and batch padding.

MXNet-lineage role (README.md:23,53-56): `resize(im, target, max)` = cv2.resize(fx=fy=scale, INTER_LINEAR) on the host,
`transform(im, pixel_means)` = BGR->RGB, minus mean, HWC->CHW, then `tensor_vstack` zero-pads the batch. Here the host
only computes the scale and moves the 8-bit frame; flip + resize + normalise + layout + padding are ONE kernel launch
per batch (csrc/preprocess.hip) writing the bf16 NCHW tensor the stem kernel reads. The box / polygon coordinate
transforms are a few floats per object and stay on the host (numpy).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import ImageDescT, check, ptr, stream_ptr

PIXEL_MEANS = (123.68, 116.779, 103.939)      # RGB ImageNet means (lineage config default), std 1
PIXEL_STDS = (1.0, 1.0, 1.0)


def resize_scale(h, w, target_size=800, max_size=1333):
    """Shorter side to target_size unless that pushes the longer side past max_size (lineage `resize`)."""
    lo, hi = (h, w) if h < w else (w, h)
    scale = float(target_size) / float(lo)
    if np.round(scale * hi) > max_size:
        scale = float(max_size) / float(hi)
    return scale


def resized_shape(h, w, scale):
    """cv2.resize(fx=fy=scale) output size: saturate_cast<int>(src * scale) = round half to even."""
    return int(round(h * scale)), int(round(w * scale))


def pad_shape(shapes, multiple=32):
    """Common (Hp, Wp) of a batch, each rounded up to `multiple` (FPN needs /32)."""
    hp = max(s[0] for s in shapes)
    wp = max(s[1] for s in shapes)
    return (hp + multiple - 1) // multiple * multiple, (wp + multiple - 1) // multiple * multiple


def flip_boxes(boxes, width):
    """Mirror [G,4+] pixel boxes (inclusive corners) in an image of `width` columns: x1' = w - x2 - 1, x2' = w - x1 - 1."""
    out = np.array(boxes, dtype=np.float32, copy=True)
    if out.size:
        x1 = out[:, 0].copy()
        out[:, 0] = width - out[:, 2] - 1
        out[:, 2] = width - x1 - 1
    return out


def transform_boxes(boxes, scale, flip, width):
    """Source-image boxes -> network-input boxes: flip in the source frame, then scale (the order the lineage uses:
    the roidb is flipped, the loader scales)."""
    b = flip_boxes(boxes, width) if flip else np.array(boxes, dtype=np.float32, copy=True)
    if b.size:
        b[:, :4] *= np.float32(scale)
    return b


def transform_polygons(polys, scale, flip, width):
    """polys: list of flat [x0,y0,x1,y1,...] lists in source pixels (COCO convention: continuous coordinates).
    Flip maps x -> width - x (continuous), then scale."""
    out = []
    for p in polys:
        a = np.asarray(p, dtype=np.float32).reshape(-1, 2).copy()
        if flip:
            a[:, 0] = np.float32(width) - a[:, 0]
        out.append(a * np.float32(scale))
    return out


def pack_polygons(per_instance_polys, N, G):
    """per_instance_polys[n][g] = list of [V,2] arrays -> (verts [V,2] f32, poly_start [P+1] i32, inst_first [N*G+1] i32)."""
    verts, poly_start, inst_first = [], [0], [0]
    for n in range(N):
        row = per_instance_polys[n] if n < len(per_instance_polys) else []
        for g in range(G):
            for poly in (row[g] if g < len(row) else []):
                a = np.asarray(poly, dtype=np.float32).reshape(-1, 2)
                verts.append(a)
                poly_start.append(poly_start[-1] + a.shape[0])
            inst_first.append(len(poly_start) - 1)
    v = np.concatenate(verts, axis=0) if verts else np.zeros((0, 2), np.float32)
    return np.ascontiguousarray(v, np.float32), np.asarray(poly_start, np.int32), np.asarray(inst_first, np.int32)


def polygon_masks(verts, poly_start, inst_first, N, G, H, W, out=None):
    """Device tensors in, [N,G,H,W] u8 instance masks out (mxdet_polygon_masks)."""
    import torch
    lib = _lib.load()
    if out is None:
        out = torch.empty((N, G, H, W), dtype=torch.uint8, device=poly_start.device)
    check(lib.mxdet_polygon_masks(ptr(verts), ptr(poly_start), ptr(inst_first), N, G, H, W, ptr(out), stream_ptr()),
          "polygon_masks")
    return out


class BatchPreprocessor:
    """Decoded u8 frames (already on the device) -> normalised, padded bf16 NCHW batch in one launch."""

    def __init__(self, target_size=800, max_size=1333, means=PIXEL_MEANS, stds=PIXEL_STDS, swap_rb=False,
                 pad_to=None, multiple=32, fit_inside=False):
        self.target_size, self.max_size, self.fit_inside = target_size, max_size, fit_inside
        self.means = (C.c_float * 3)(*means)
        self.stds = (C.c_float * 3)(*stds)
        self.swap_rb, self.pad_to, self.multiple = int(swap_rb), pad_to, multiple

    def plan(self, shapes):
        """[(h, w)] -> (scales, resized shapes, (Hp, Wp)). pad_to: None = smallest /multiple shape holding the batch,
        (Hp, Wp) = fixed (with fit_inside: frames are scaled down to fit it), "orient" = (800,1344)-style landscape or
        its transpose for portrait batches."""
        scales = [resize_scale(h, w, self.target_size, self.max_size) for (h, w) in shapes]
        if self.fit_inside and isinstance(self.pad_to, tuple):
            # one fixed batch shape for every frame (a single captured hipGraph): frames of the other orientation are
            # scaled down until they fit -- a deviation from the lineage, which pads each batch to its own shape
            hp, wp = self.pad_to
            for n, (h, w) in enumerate(shapes):
                s = min(scales[n], hp / float(h), wp / float(w))
                while s > 0 and (int(round(h * s)) > hp or int(round(w * s)) > wp):
                    s = s * (1.0 - 1e-6)
                scales[n] = s
        rs = [resized_shape(h, w, s) for (h, w), s in zip(shapes, scales)]
        if self.pad_to == "orient":
            # two fixed shapes (landscape / portrait) so that a captured hipGraph per orientation can be replayed;
            # aspect-grouped batches never mix them, a mixed batch gets the square that holds both
            m = self.multiple
            lo, hi = (self.target_size + m - 1) // m * m, (self.max_size + m - 1) // m * m
            land = all(w >= h for (h, w) in shapes)
            port = all(h > w for (h, w) in shapes)
            hp, wp = (lo, hi) if land else (hi, lo) if port else (hi, hi)
        else:
            hp, wp = self.pad_to if self.pad_to is not None else pad_shape(rs, self.multiple)
        wp = (wp + 7) // 8 * 8
        return scales, rs, (hp, wp)

    def __call__(self, frames, flips=None, out=None):
        """frames: list of device u8 tensors [h,w,3] (contiguous). Returns (batch bf16 [N,3,Hp,Wp], im_info [N,3] numpy,
        scales)."""
        import torch
        lib = _lib.load()
        N = len(frames)
        flips = [False] * N if flips is None else flips
        shapes = [(int(f.shape[0]), int(f.shape[1])) for f in frames]
        scales, rs, (hp, wp) = self.plan(shapes)
        if out is None:
            out = torch.empty((N, 3, hp, wp), dtype=torch.bfloat16, device=frames[0].device)
        descs = (ImageDescT * max(N, 1))()
        for n, f in enumerate(frames):
            assert f.dtype == torch.uint8 and f.is_contiguous() and f.shape[2] == 3
            d = descs[n]
            d.src, d.src_h, d.src_w = f.data_ptr(), shapes[n][0], shapes[n][1]
            d.dst_h, d.dst_w, d.flip, d.inv_scale = rs[n][0], rs[n][1], int(bool(flips[n])), 1.0 / scales[n]
        check(lib.mxdet_image_preprocess(C.cast(descs, C.c_void_p), N, hp, wp, C.cast(self.means, C.c_void_p),
                                         C.cast(self.stds, C.c_void_p), self.swap_rb, ptr(out), stream_ptr()),
              "image_preprocess")
        im_info = np.array([[rs[n][0], rs[n][1], scales[n]] for n in range(N)], np.float32).reshape(N, 3)
        return out, im_info, scales
