"""What a plain streaming kernel gets at the tensor sizes of the step's 1x1 layers (warm, hipGraph of 10 re-issues):
out = a + b (bf16, 16 B per lane): 3 tensors of `mb` MB each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mxdetection_amd.ops import dense


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for elems in (2100 * 2048, 8400 * 1024, 33600 * 512, 134400 * 256, 4 * 134400 * 256):
    a = torch.randn(elems, device="cuda").to(torch.bfloat16)
    b = torch.randn(elems, device="cuda").to(torch.bfloat16)
    o = torch.empty_like(a)
    t = timeit(lambda: dense.add_bf16(a, b, o))
    print("%7.1f MB per tensor: %6.1f us  %.2f TB/s" % (elems * 2 / 1e6, t, 3 * elems * 2 / t / 1e6), flush=True)
