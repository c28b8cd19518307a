#!/usr/bin/env python
"""Per-layer-shape timing of the conv families inside the real training step (HIP events around every launch).

  python tools/conv_report.py [--steps 3]      (GPU box)
Prints one line per (family, shape): launches/step, avg us, TFLOP/s, share of the step's conv time.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from mxdetection_amd.models import FasterRCNN
    from mxdetection_amd.ops import dense
    recs = []
    on = [False]

    def wrap(fn, family, key_of):
        def inner(*a, **kw):
            if not on[0]:
                return fn(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **kw)
            e.record()
            recs.append((family, key_of(*a, **kw), s, e))
            return out
        return inner

    def k_fwd(x, w, bias=None, residual=None, stride=1, pad=0, relu=False, res_upsample=False, out=None):
        N, H, W, Cin = x.shape
        Cout, KH, KW, _ = w.shape
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        return (N * Ho * Wo, Cout, KH * KW * Cin, KH, stride)

    def k_dgrad(dy, wt, x_shape, KH, KW, stride=1, pad=0, residual=None, relu_mask=None, accumulate=False, out=None):
        N, Ho, Wo, Cout = dy.shape
        return (x_shape[0] * x_shape[1] * x_shape[2], x_shape[3], KH * KW * Cout, KH, stride)

    def k_wgrad(x, dy, KH, KW, stride=1, pad=0, dw=None, db=None, accumulate=False, workspace=None):
        N, Ho, Wo, Cout = dy.shape
        return (Cout, KH * KW * x.shape[3], N * Ho * Wo, KH, stride)

    dense.conv2d_forward = wrap(dense.conv2d_forward, "fwd", k_fwd)
    dense.conv2d_dgrad = wrap(dense.conv2d_dgrad, "dgrad", k_dgrad)
    dense.conv2d_wgrad = wrap(dense.conv2d_wgrad, "wgrad", k_wgrad)
    m = FasterRCNN("cuda", seed=7)
    batch = bench.synth_batch(0, 0, "cuda")
    for i in range(2):
        m.train_step(*batch, step=i)
    torch.cuda.synchronize()
    on[0] = True
    for i in range(args.steps):
        m.train_step(*batch, step=2 + i)
    torch.cuda.synchronize()
    agg = {}
    for fam, key, s, e in recs:
        a = agg.setdefault((fam, key), [0.0, 0])
        a[0] += s.elapsed_time(e) * 1e-3
        a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print("%-6s %9s %6s %7s k s  %5s %9s %8s %6s" % ("family", "M", "N", "K", "n/st", "avg_us", "TFLOP/s", "share"))
    for (fam, key), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        M, Nn, K, k, s = key
        fl = 2.0 * M * Nn * K * n
        print("%-6s %9d %6d %7d %d %d  %5.1f %9.1f %8.1f %5.1f%%" % (fam, M, Nn, K, k, s, n / args.steps, 1e6 * t / n,
                                                                   fl / t / 1e12, 100 * t / tot))
    print("conv total per step: %.3f ms" % (1e3 * tot / args.steps))


if __name__ == "__main__":
    main()
