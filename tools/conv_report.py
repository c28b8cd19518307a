#!/usr/bin/env python
"""Per-layer-shape timing of the conv families of the real training step, GPU-saturated (each logged launch is
re-issued REPS times back to back between HIP events).   python tools/conv_report.py   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from mxdetection_amd.models import FasterRCNN
    timer = bench.ConvTimer()
    timer.install()
    m = FasterRCNN("cuda", seed=7)
    m.enable_grouped_wgrad()
    batch = bench.synth_batch(0, 0, "cuda")
    for i in range(2):
        m.train_step(*batch, step=i, lr=1e-4)
    timer.logging = True
    m.forward_backward(*batch, step=5)
    timer.logging = False
    torch.cuda.synchronize()
    agg = {}
    REPS = 8
    for family, flops, fn, a, kw, _shape in timer.log:
        t = bench.ConvTimer.time_launch(fn, a, kw, REPS)
        plan = a[0] if a and hasattr(a[0], "grid") or (a and hasattr(a[0], "grid_wgrad")) else None
        if plan is not None:     # grouped launch: (layers in the group, -, -, 0, 0)
            key = (plan.n, 0, 0, 0, 0)
            family = family + "_grouped"
        elif "chained" in _shape:   # conv2d_forward_chain: 3x3 -> 1x1 (-> next 1x1) in one launch; N column = the 1x1's columns
            x, w, w2 = a[0], a[1], a[3]
            key = (x.shape[0] * x.shape[1] * x.shape[2], w2.shape[0], w.shape[1] * w.shape[2] * w.shape[3], 3, 1)
            family = family + "_chain"
        elif family == "conv_igemm_fwd":
            x, w = a[0], a[1]
            stride = kw.get("stride", a[4] if len(a) > 4 else 1)
            key = (x.shape[0] * ((x.shape[1] - 1) // stride + 1) * ((x.shape[2] - 1) // stride + 1), w.shape[0],
                   w.shape[1] * w.shape[2] * w.shape[3], w.shape[1], stride)
        elif family == "conv_igemm_dgrad":
            dy, wt, xs = a[0], a[1], a[2]
            key = (xs[0] * xs[1] * xs[2], xs[3], wt.shape[1] * wt.shape[2] * wt.shape[3], wt.shape[1], a[5] if len(a) > 5 else kw.get("stride", 1))
        else:
            x, dy = a[0], a[1]
            key = (dy.shape[3], a[2] * a[3] * x.shape[3], dy.shape[0] * dy.shape[1] * dy.shape[2], a[2], a[4] if len(a) > 4 else kw.get("stride", 1))
        r = agg.setdefault((family[5:] + ("" if plan is None else "#%d" % id(plan)), key), [0.0, 0, 0.0])
        r[0] += t
        r[1] += 1
        r[2] += flops
    tot = sum(v[0] for v in agg.values())
    print("grouped launches: M column = number of layers in the group")
    print("%-11s %8s %6s %7s k s  %4s %9s %8s %6s" % ("family", "M", "N", "K", "n", "avg_us", "TFLOP/s", "share"))
    for (fam, key), (t, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        M, Nn, K, k, s = key
        print("%-11s %8d %6d %7d %d %d  %4d %9.1f %8.1f %5.1f%%" % (fam.split("#")[0][:20], M, Nn, K, k, s, n, 1e6 * t / n, fl / t / 1e12, 100 * t / tot))
    print("conv total per step: %.3f ms" % (1e3 * tot))


if __name__ == "__main__":
    main()
