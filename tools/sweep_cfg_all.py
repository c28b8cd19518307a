"""Tile-configuration sweep over the (ungrouped) conv shapes of the R50-FPN step: heuristic vs every forced config,
fwd and dgrad, graph-timed in one process.  python tools/sweep_cfg_all.py > gpurun_out/sweep_cfg_all.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense
lib = _lib.load()
CFGS = [int(c) for c in os.environ.get("SWEEP_CFGS", "40,41,45,46,49").split(",")]
STATIC_ONLY = any(40 <= c < 50 or 60 <= c < 70 for c in CFGS)     # cfgs 40-49 assume a stride-1 1x1 / 3x3 layer
# (H, W, Cin, Cout, k, stride) at N = 2
SHAPES = [(200, 336, 64, 64, 1, 1), (200, 336, 64, 64, 3, 1), (200, 336, 64, 256, 1, 1), (200, 336, 256, 64, 1, 1),
          (200, 336, 256, 128, 1, 1), (200, 336, 128, 128, 3, 2), (100, 168, 128, 512, 1, 1), (100, 168, 512, 128, 1, 1),
          (100, 168, 128, 128, 3, 1), (200, 336, 256, 512, 1, 2), (100, 168, 512, 256, 1, 1), (100, 168, 256, 256, 3, 2),
          (50, 84, 256, 1024, 1, 1), (50, 84, 1024, 256, 1, 1), (50, 84, 256, 256, 3, 1), (100, 168, 512, 1024, 1, 2),
          (50, 84, 1024, 512, 1, 1), (50, 84, 512, 512, 3, 2), (25, 42, 512, 2048, 1, 1), (25, 42, 2048, 512, 1, 1),
          (25, 42, 512, 512, 3, 1), (50, 84, 1024, 2048, 1, 2), (200, 336, 256, 256, 1, 1), (100, 168, 512, 256, 1, 1),
          (50, 84, 1024, 256, 1, 1), (25, 42, 2048, 256, 1, 1), (200, 336, 256, 64, 1, 1)]


def timeit(fn, reps=12):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


torch.manual_seed(0)
tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0]}
for (H, W, Cin, Cout, K, s) in SHAPES:
    if os.environ.get("SWEEP_MAXH") and H > int(os.environ["SWEEP_MAXH"]):
        continue
    p = K // 2
    x = torch.randn(2, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
    wt = dense.filter_transpose(w)
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    y = torch.empty(2, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(2, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty_like(x)
    bias = torch.randn(Cout, device="cuda")
    for kind in ("fwd", "dgrad"):
        if kind == "dgrad" and (Cout % 64 or s == 2):
            continue          # stride-2 dgrad has its own (parity) path
        if STATIC_ONLY and s != 1:
            continue
        fn = (lambda: dense.conv2d_forward(x, w, bias, None, s, p, True, False, y)) if kind == "fwd" else \
             (lambda: dense.conv2d_dgrad(dy, wt, tuple(x.shape), K, K, s, p, relu_mask=x, out=dx))
        lib.mxdet_debug_force_conv_cfg(0)
        t0 = timeit(fn)
        res = {}
        for c in CFGS:
            lib.mxdet_debug_force_conv_cfg(c)
            try:
                res[c] = timeit(fn)
            except Exception:  # noqa: BLE001
                res[c] = float("inf")
        lib.mxdet_debug_force_conv_cfg(0)
        best = min(res, key=res.get)
        tot[kind][0] += t0
        tot[kind][1] += min(t0, res[best])
        print("%-5s %3dx%-3d %4d->%-4d k%d s%d: heur %6.1f us | best cfg%-2d %6.1f us (%+5.1f%%) | %s" % (
            kind, H, W, Cin, Cout, K, s, t0, best, res[best], 100.0 * (res[best] - t0) / t0,
            " ".join("%d:%.1f" % (c, res[c]) for c in CFGS)), flush=True)
for k, (a, b) in tot.items():
    print("%s total: heuristic %.1f us, per-shape best %.1f us (%.1f%%)" % (k, a, b, 100.0 * (b - a) / a))
