"""Compare forced conv tile configurations against the heuristic one (bitwise for the same K order is not expected
across split-K variants: tolerance 2^-7 like the dense tests) and time them.  python tools/check_cfg.py cfg [cfg ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense
lib = _lib.load()
cfgs = [int(c) for c in sys.argv[1:]] or [20, 22]
shapes = [(2, 50, 84, 256, 256, 3, 1), (2, 25, 42, 512, 512, 3, 1), (2, 25, 42, 2048, 512, 1, 1), (2, 25, 42, 512, 2048, 1, 1),
          (2, 50, 84, 1024, 256, 1, 1), (2, 50, 84, 256, 1024, 1, 1), (1, 13, 21, 192, 200, 3, 1), (2, 100, 168, 128, 128, 3, 1),
          (2, 200, 336, 256, 256, 3, 1), (2, 200, 336, 64, 256, 1, 1), (2, 100, 168, 512, 128, 1, 1)]
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
torch.manual_seed(0)
for (N, H, W, Cin, Cout, K, s) in shapes:
    p = K // 2
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(N, H, W, Cout, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, W, Cout, device="cuda").to(torch.bfloat16)
    wt = dense.filter_transpose(w)
    lib.mxdet_debug_force_conv_cfg(0)
    y0 = dense.conv2d_forward(x, w, bias, res, s, p, True).float()
    d0 = dense.conv2d_dgrad(dy, wt, tuple(x.shape), K, K, s, p, relu_mask=x).float() if Cout % 64 == 0 else None
    t0 = timeit(lambda: dense.conv2d_forward(x, w, bias, res, s, p, True, False, torch.empty_like(res)))
    line = "N%d %dx%d %d->%d k%d: heur %.1f us" % (N, H, W, Cin, Cout, K, t0)
    for c in cfgs:
        lib.mxdet_debug_force_conv_cfg(c)
        y = dense.conv2d_forward(x, w, bias, res, s, p, True).float()
        err = ((y - y0).abs() / (y0.abs() * 2 ** -7 + y0.pow(2).mean().sqrt() * 2 ** -7)).max().item()
        derr = 0.0
        if d0 is not None:
            d = dense.conv2d_dgrad(dy, wt, tuple(x.shape), K, K, s, p, relu_mask=x).float()
            derr = ((d - d0).abs() / (d0.abs() * 2 ** -7 + d0.pow(2).mean().sqrt() * 2 ** -7)).max().item()
        t = timeit(lambda: dense.conv2d_forward(x, w, bias, res, s, p, True, False, torch.empty_like(res)))
        line += " | cfg%d %.1f us (err %.2f/%.2f)" % (c, t, err, derr)
    print(line)
lib.mxdet_debug_force_conv_cfg(0)
