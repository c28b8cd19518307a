"""CPU restatement of the whole Faster R-CNN R50-FPN training step. TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED by the reference (/root/reference holds no code: see mxdet_oracle.c). Dense arithmetic is
torch-CPU fp32 autograd (an independent implementation of conv / linear / pooling); every detection op
(anchors, anchor targets, proposals + NMS, proposal targets) is the C oracle; RoIAlign is restated here in
differentiable torch ops with the same sampling geometry. Used by
  * tests/test_gpu_model_parity.py : losses and gradients of the HIP step vs this model (same weights)
  * bench.py cpu_baseline          : the "CPU path timed on the same host" leg
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

STRIDES = [4, 8, 16, 32, 64]
BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def _nchw(w):
    """[Cout,KH,KW,Cin] -> [Cout,Cin,KH,KW]"""
    return w.permute(0, 3, 1, 2).contiguous()


def random_params(depth=50, seed=7, num_classes=81, head_cpad=64):
    """Same layer set / shapes as mxdetection_amd.models.FasterRCNN (values are independent random draws)."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, cin, cout, k, std=None, bias=True):
        std = math.sqrt(2.0 / (k * k * cin)) if std is None else std
        p[name + ".weight"] = torch.randn((cout, k, k, cin), generator=g) * std
        if bias:
            p[name + ".bias"] = torch.zeros((cout,))

    conv("stem", 3, 64, 7)
    cins = [64, 256, 512, 1024]
    for si, nb in enumerate(BLOCKS[depth]):
        planes = 64 << si
        for bi in range(nb):
            cin = cins[si] if bi == 0 else planes * 4
            n = "layer%d.%d" % (si + 1, bi)
            conv(n + ".conv1", cin, planes, 1)
            conv(n + ".conv2", planes, planes, 3)
            conv(n + ".conv3", planes, planes * 4, 1, std=0.25 * math.sqrt(2.0 / planes))
            if bi == 0:
                conv(n + ".down", cin, planes * 4, 1)
    for i, c in enumerate([256, 512, 1024, 2048]):
        conv("fpn.lat%d" % (i + 2), c, 256, 1, std=math.sqrt(1.0 / c))
        conv("fpn.out%d" % (i + 2), 256, 256, 3, std=math.sqrt(1.0 / (9 * 256)))
    conv("rpn.conv", 256, 256, 3, std=0.01)
    conv("rpn.out", 256, head_cpad, 1, std=0.01)
    ld = (num_classes * 5 + 63) // 64 * 64
    conv("bbox.fc1", 7 * 7 * 256, 1024, 1)
    conv("bbox.fc2", 1024, 1024, 1)
    conv("bbox.fc_out", 1024, ld, 1, std=0.01)
    return p


def roi_align_torch(feats, scales, rois, levels, PH=7, PW=7, sr=2, lvl_min=2):
    """feats[l]: [N,C,H,W] fp32 (requires_grad ok). Same geometry as oracle_roi_align (aligned=False)."""
    R = rois.shape[0]
    C = feats[0].shape[1]
    out = torch.zeros((R, C, PH, PW), dtype=feats[0].dtype)
    for l, f in enumerate(feats):
        idx = torch.nonzero(levels == l + lvl_min).flatten()
        if idx.numel() == 0:
            continue
        r = rois[idx]
        N, _, H, W = f.shape
        s = scales[l]
        b = r[:, 0].long().clamp(0, N - 1)
        sw, sh, ew, eh = r[:, 1] * s, r[:, 2] * s, r[:, 3] * s, r[:, 4] * s
        rw, rh = (ew - sw).clamp(min=1.0), (eh - sh).clamp(min=1.0)
        bh, bw = rh / PH, rw / PW
        ph = torch.arange(PH, dtype=torch.float32)
        iy = torch.arange(sr, dtype=torch.float32)
        # y[r, ph, iy], x[r, pw, ix]
        y = sh[:, None, None] + ph[None, :, None] * bh[:, None, None] + ((iy[None, None, :] + 0.5) * bh[:, None, None]) / sr
        x = sw[:, None, None] + ph[None, :, None] * bw[:, None, None] + ((iy[None, None, :] + 0.5) * bw[:, None, None]) / sr
        vy = ~((y < -1.0) | (y > H))
        vx = ~((x < -1.0) | (x > W))
        y, x = y.clamp(min=0.0), x.clamp(min=0.0)
        yl, xl = y.floor().long(), x.floor().long()
        top, right = yl >= H - 1, xl >= W - 1
        yl, xl = torch.where(top, torch.full_like(yl, H - 1), yl), torch.where(right, torch.full_like(xl, W - 1), xl)
        yh, xh = torch.where(top, yl, yl + 1), torch.where(right, xl, xl + 1)
        y, x = torch.where(top, yl.float(), y), torch.where(right, xl.float(), x)
        ly, lx = y - yl.float(), x - xl.float()
        hy, hx = 1.0 - ly, 1.0 - lx
        fl = f.permute(0, 2, 3, 1)   # [N,H,W,C]

        def tap(yy, xx, wy, wx):
            # -> [r, PH, sr, PW, sr, C]
            v = fl[b[:, None, None, None, None], yy[:, :, :, None, None], xx[:, None, None, :, :]]
            w = (wy * vy.float())[:, :, :, None, None] * (wx * vx.float())[:, None, None, :, :]
            return v * w[..., None]

        acc = tap(yl, xl, hy, hx) + tap(yl, xh, hy, lx) + tap(yh, xl, ly, hx) + tap(yh, xh, ly, lx)
        pooled = acc.sum(dim=(2, 4)) / float(sr * sr)          # [r, PH, PW, C]
        out = out.index_put((idx,), pooled.permute(0, 3, 1, 2))
    return out


class RefModel:
    def __init__(self, params, depth=50, num_classes=81, A=3, rois_per_image=512, pre_n=2000, post_n=2000, seed=99):
        self.p = {k: v.clone().float().requires_grad_(True) for k, v in params.items()}
        self.depth, self.nc, self.A, self.R, self.pre_n, self.post_n, self.seed = depth, num_classes, A, rois_per_image, pre_n, post_n, seed
        self.base = [O.base_anchors(s) for s in STRIDES]

    def conv(self, name, x, stride=1, pad=0, relu=False):
        y = F.conv2d(x, _nchw(self.p[name + ".weight"]), self.p.get(name + ".bias"), stride=stride, padding=pad)
        return F.relu(y) if relu else y

    def features(self, image):
        x = F.relu(F.conv2d(image, _nchw(self.p["stem.weight"]), self.p["stem.bias"], stride=2, padding=3))
        x = F.max_pool2d(x, 3, 2, 1)
        C = []
        for si, nb in enumerate(BLOCKS[self.depth]):
            for bi in range(nb):
                n = "layer%d.%d" % (si + 1, bi)
                s = 2 if (bi == 0 and si > 0) else 1
                a = self.conv(n + ".conv1", x, relu=True)
                a = self.conv(n + ".conv2", a, stride=s, pad=1, relu=True)
                sc = self.conv(n + ".down", x, stride=s) if bi == 0 else x
                x = F.relu(self.conv(n + ".conv3", a) + sc)
            C.append(x)
        inner = [None] * 4
        for i in (3, 2, 1, 0):
            inner[i] = self.conv("fpn.lat%d" % (i + 2), C[i])
            if i < 3:
                up = inner[i + 1].repeat_interleave(2, 2).repeat_interleave(2, 3)[:, :, :inner[i].shape[2], :inner[i].shape[3]]
                inner[i] = inner[i] + up
        P = [self.conv("fpn.out%d" % (i + 2), inner[i], pad=1) for i in range(4)]
        P.append(P[3][:, :, ::2, ::2])
        return C, P

    def rpn(self, P):
        hs = []
        for p in P:
            t = self.conv("rpn.conv", p, pad=1, relu=True)
            hs.append(self.conv("rpn.out", t))       # [N,Cpad,H,W]
        return hs

    def step(self, image, gt, im_info, step=0, image_offset=0, forced_rois=None):
        """image [N,3,H,W] fp32; gt [N,G,5] numpy; returns dict(losses, grads, rois, ...)."""
        N = image.shape[0]
        A = self.A
        _, P = self.features(image)
        hs = self.rpn(P)
        shapes = [(h.shape[2], h.shape[3]) for h in hs]
        anchors = np.concatenate([O.grid_anchors(self.base[l], H, W, STRIDES[l]) for l, (H, W) in enumerate(shapes)])
        labels, matched, targets, _ = O.anchor_target(anchors, gt, im_info, 0.7, 0.3, 0.0, 256, 0.5, self.seed, step,
                                                      image_offset)
        # canonical (level, y, x, a) order of the head outputs
        logit = torch.cat([h[:, :A].permute(0, 2, 3, 1).reshape(N, -1) for h in hs], 1)
        delta = torch.cat([h[:, A:5 * A].permute(0, 2, 3, 1).reshape(N, -1, 4) for h in hs], 1)
        lab = torch.from_numpy(labels)
        tgt = torch.from_numpy(targets)
        norm = 1.0 / (N * 256)
        valid = lab >= 0
        rpn_cls = F.binary_cross_entropy_with_logits(logit[valid], lab[valid].float(), reduction="sum") * norm
        fg = lab == 1
        d = (delta[fg] - tgt[fg]) * 9.0   # sigma^2 * x
        ad = (delta[fg] - tgt[fg]).abs()
        rpn_reg = torch.where(ad < 1.0 / 9.0, 0.5 * d * (delta[fg] - tgt[fg]), ad - 0.5 / 9.0).sum() * norm
        # proposals from this model's own head outputs (or forced, for teacher-forced parity)
        if forced_rois is None:
            sc = [h[:, :A].permute(0, 2, 3, 1).reshape(N, -1).detach().numpy() for h in hs]
            dl = [h[:, A:5 * A].permute(0, 2, 3, 1).reshape(N, -1, 4).detach().numpy() for h in hs]
            rois, _, _, num = O.proposal(sc, dl, self.base, [s[0] for s in shapes], [s[1] for s in shapes], STRIDES,
                                         im_info, self.pre_n, self.post_n, 0.7, 0.0)
        else:
            rois, num = forced_rois
        srois, slab, stgt, swgt, _, nfg = O.proposal_target(rois, num, gt, self.R, 0.25, 0.5, 0.5, 0.0, self.nc, False,
                                                            (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), self.seed, step,
                                                            image_offset)
        r2 = torch.from_numpy(srois.reshape(-1, 5))
        levels = torch.from_numpy(O.fpn_level(srois.reshape(-1, 5)))
        pooled = roi_align_torch(P[:4], [1.0 / s for s in STRIDES[:4]], r2, levels)      # [R,C,7,7]
        x = pooled.permute(0, 2, 3, 1).reshape(pooled.shape[0], -1)                       # (ph,pw,c) order
        h1 = F.relu(F.linear(x, self.p["bbox.fc1.weight"].reshape(1024, -1), self.p["bbox.fc1.bias"]))
        h2 = F.relu(F.linear(h1, self.p["bbox.fc2.weight"].reshape(1024, -1), self.p["bbox.fc2.bias"]))
        o = F.linear(h2, self.p["bbox.fc_out.weight"].reshape(-1, 1024), self.p["bbox.fc_out.bias"])
        Rt = o.shape[0]
        sl = torch.from_numpy(slab.reshape(-1)).long()
        ok = sl >= 0
        rcnn_cls = F.cross_entropy(o[ok][:, :self.nc], sl[ok], reduction="sum") / Rt
        dd = (o[:, self.nc:self.nc * 5] - torch.from_numpy(stgt.reshape(Rt, -1))) * torch.from_numpy(swgt.reshape(Rt, -1))
        rcnn_reg = torch.where(dd.abs() < 1.0, 0.5 * dd * dd, dd.abs() - 0.5)[sl > 0].sum() / Rt
        total = rpn_cls + rpn_reg + rcnn_cls + rcnn_reg
        for v in self.p.values():
            v.grad = None
        total.backward()
        grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in self.p.items()}
        return {"losses": [float(v.detach()) for v in (rpn_cls, rpn_reg, rcnn_cls, rcnn_reg)], "grads": grads,
                "rois": rois, "num_rois": num, "sampled_rois": srois, "labels": slab, "heads": hs}


def timed_cpu_baseline(threads=None):
    """One full-size image (3x800x1333 padded to 1344), forward + targets + losses + backward on the host CPU."""
    import os
    n_cpu = os.cpu_count() or 1
    threads = threads or min(n_cpu, 64)
    torch.set_num_threads(threads)
    params = random_params()
    m = RefModel(params)
    rng = np.random.default_rng(4321)
    gt = -np.ones((1, 100, 5), np.float32)
    for k in range(8):
        w, h = float(np.exp(rng.uniform(np.log(16), np.log(600)))), float(np.exp(rng.uniform(np.log(16), np.log(600))))
        w, h = min(w, 1332.0), min(h, 799.0)
        x1, y1 = float(rng.uniform(0, 1333 - w)), float(rng.uniform(0, 800 - h))
        gt[0, k] = [x1, y1, x1 + w - 1, y1 + h - 1, float(rng.integers(1, 81))]
    info = np.array([[800, 1333, 1.0]], np.float32)
    g = torch.Generator().manual_seed(1234)
    warm = torch.randn((1, 3, 128, 160), generator=g)
    m.step(warm, gt * 0 - 1 + 0, np.array([[128, 160, 1.0]], np.float32))     # warm-up: load kernels, no GT
    img = torch.zeros((1, 3, 800, 1344))
    img[..., :1333] = torch.randn((1, 3, 800, 1333), generator=g)
    # bounded sample: full-size steps until ~12 s of CPU work have been timed (at least 2, at most 6 images)
    times = []
    while len(times) < 2 or (sum(times) < 12.0 and len(times) < 6):
        t0 = time.perf_counter()
        out = m.step(img, gt, info)
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d images 3x800x1333 (padded 1344), one full train step each (fwd+targets+losses+bwd), %.1f s of "
                      "CPU work, mean %.2f s/image; own CPU oracle (torch-CPU fp32 dense + C detection ops), not MXNet "
                      "(unavailable offline); host has %d cpus" % (len(times), sum(times), dt, n_cpu),
            "losses": out["losses"]}
