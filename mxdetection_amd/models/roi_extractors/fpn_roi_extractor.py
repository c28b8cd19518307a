"""FPN RoI extractor: level = clamp(floor(4 + log2(sqrt(wh)/224)), 2, 5), then one multi-level RoIAlign launch.

Plugin slot: models/roi_extractors (/root/reference/README.md:32); MXNet role contrib.ROIAlign (README.md:37).
"""
import torch

from ...ops import dense
from ...ops.roi_align import (fpn_level_map, roi_align_backward, roi_align_backward_gather,
                              roi_align_backward_gather_prepare, roi_align_backward_gather_workspace, roi_align_forward)


class FPNRoIExtractor:
    def __init__(self, strides, pooled=(7, 7), sampling_ratio=2, lvl_min=2, device="cuda"):
        self.scales = [1.0 / s for s in strides]
        self.pooled, self.sr, self.lvl_min = pooled, sampling_ratio, lvl_min
        self.lvl_max = lvl_min + len(strides) - 1
        self.device = device
        self.out = None
        self.outs = {}
        self.dacc = None
        # forward(prepare_gather=True), i.e. training with the gather-form backward: the per-roi records of the gather are
        # written right there, in the forward pass (they depend only on the rois), into a workspace of this extractor's own
        self.gather_ws = {}
        self.prepared = None

    def forward(self, feats, rois, prepare_gather=False):
        """feats: P2..P5 (bf16 [N,H,W,C]); rois [R,5] f32."""
        self.feats, self.rois = feats[:len(self.scales)], rois
        self.levels = fpn_level_map(rois, self.lvl_min, self.lvl_max)
        R, C = rois.shape[0], feats[0].shape[3]
        # keyed by shape: a call with another roi count (inference) must not free the buffer a captured step writes
        self.out = self.outs.setdefault((R, C), None)
        if self.out is None:
            self.out = torch.empty((R, self.pooled[0], self.pooled[1], C), dtype=torch.bfloat16, device=self.device)
            self.outs[(R, C)] = self.out
        self.prepared = None
        if prepare_gather:
            # the workspace size depends on the roi count AND on the pyramid geometry (per-row roi lists): a portrait batch
            # after a landscape one needs a larger one. Buffers are kept per key, never freed (a captured step may hold one)
            key = (R,) + tuple(tuple(f.shape[:3]) for f in self.feats)
            ws = self.gather_ws.get(key)
            if ws is None:
                ws = self.gather_ws[key] = roi_align_backward_gather_workspace(self.feats, self.scales, R, self.lvl_min)
            roi_align_backward_gather_prepare(self.feats, self.scales, rois, self.levels, self.pooled, self.sr,
                                              self.lvl_min, ws)
            self.prepared = ws
        return roi_align_forward(self.feats, self.scales, rois, self.levels, self.pooled, self.sr, self.lvl_min, self.out)

    def backward(self, grad_out, dP, shared_acc=None, zero=True, finalize=True):
        """Scatter grad_out into fp32 accumulators, then write them (rounded once) into the bf16 dP[l].
        Several extractors (box + mask branch) may share one set of accumulators: the first zeroes, the last
        finalizes."""
        if shared_acc is not None:
            self.dacc = shared_acc
        if self.dacc is None:
            # one flat fp32 buffer, the levels are views: zero-fill and finalize are ONE launch each
            sizes = [f.numel() for f in self.feats]
            self.dacc_flat = torch.empty((sum(sizes),), dtype=torch.float32, device=self.device)
            self.dacc, off = [], 0
            for f, n in zip(self.feats, sizes):
                self.dacc.append(self.dacc_flat[off:off + n].view(f.shape))
                off += n
        if zero:
            if getattr(self, "dacc_flat", None) is not None:
                self.dacc_flat.zero_()
            else:
                for a in self.dacc:
                    a.zero_()
        roi_align_backward(self.dacc, self.scales, self.rois, self.levels, grad_out, self.sr, self.lvl_min)
        if finalize:
            self.finalize(dP)
        return self.dacc

    def backward_gather(self, grad_out, dP, accumulate=True):
        """Deterministic gather form: adds this extractor's gradient straight into the bf16 maps dP[l] (no fp32
        accumulators, no atomics, no zero-fill / finalize passes)."""
        roi_align_backward_gather(dP, self.scales, self.rois, self.levels, grad_out, self.sr, self.lvl_min,
                                  accumulate=accumulate, workspace=self.prepared, prepared=self.prepared is not None)

    def finalize(self, dP, accumulate=False):
        """dP[l] (+)= accumulators, rounded once to bf16 (one launch when both sides are views of flat buffers laid out
        in the same order)."""
        flat = getattr(self, "dacc_flat", None)
        base = dP[0]._base if dP and dP[0]._base is not None else None
        if (flat is not None and base is not None and base.dim() == 1 and base.numel() >= flat.numel() and
                all(d._base is base for d in dP) and dP[0].storage_offset() == base.storage_offset() and
                all(dP[i + 1].storage_offset() == dP[i].storage_offset() + dP[i].numel() for i in range(len(dP) - 1))):
            dense.f32_accum_to_bf16(flat, base[:flat.numel()], accumulate=accumulate)
            return
        for a, d in zip(self.dacc, dP):
            dense.f32_accum_to_bf16(a, d, accumulate=accumulate)
