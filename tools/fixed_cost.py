"""What a small conv launch costs before its K loop does any work: graph-replayed chains of (a) the zero-fill kernel on
4 bytes (pure launch-to-launch time), (b) a one-K-step conv (K = 64) on the C4 map, (c) the same conv at real K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


z = torch.zeros(64, device="cuda")
print("tiny torch kernel (fill 64 floats): %.2f us per launch" % timeit(lambda: z.fill_(1.0)))
for (H, W, Cin, Cout, K) in [(50, 84, 64, 256, 1), (50, 84, 256, 256, 1), (50, 84, 1024, 256, 1), (50, 84, 64, 1024, 1),
                             (50, 84, 256, 1024, 1), (25, 42, 64, 512, 1), (25, 42, 2048, 512, 1), (100, 168, 64, 128, 1),
                             (100, 168, 512, 128, 1)]:
    x = torch.randn(2, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(2, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(2, H, W, Cout, device="cuda").to(torch.bfloat16)
    bias = torch.randn(Cout, device="cuda")
    t_plain = timeit(lambda: dense.conv2d_forward(x, w, None, None, 1, 0, False, False, y))
    t_full = timeit(lambda: dense.conv2d_forward(x, w, bias, res, 1, 0, True, False, y))
    mb = (x.numel() + w.numel() + y.numel()) * 2 / 1e6
    print("fwd 2x%dx%d %4d->%-4d: no epilogue operands %.1f us, bias+residual+relu %.1f us  (%.1f MB in+out -> %.1f us at 5 TB/s)" %
          (H, W, Cin, Cout, t_plain, t_full, mb, mb / 5.0))
