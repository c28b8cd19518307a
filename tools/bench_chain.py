#!/usr/bin/env python
"""The frozen C2 tail (3x3 64->64 + ReLU, 1x1 64->256 + shortcut + ReLU) as two launches vs the chained launch, graph-timed.
  python tools/bench_chain.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mxdetection_amd.ops import dense
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, W = 2, 200, 336
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((N, H, W, 64), device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn((64, 3, 3, 64), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
b = torch.randn((64,), device="cuda", generator=g)
w2 = (torch.randn((256, 1, 1, 64), device="cuda", generator=g) * 0.1).to(torch.bfloat16)
b2 = torch.randn((256,), device="cuda", generator=g)
res = torch.randn((N, H, W, 256), device="cuda", generator=g).to(torch.bfloat16)
mid = torch.empty((N, H, W, 64), device="cuda", dtype=torch.bfloat16)
y = torch.empty((N, H, W, 256), device="cuda", dtype=torch.bfloat16)
scr = torch.empty((1 << 28,), device="cuda", dtype=torch.float32)       # 1 GiB: evicts L2 and the Infinity Cache


def two():
    dense.conv2d_forward(x, w, b, None, 1, 1, True, False, mid)
    dense.conv2d_forward(mid, w2, b2, res, 1, 0, True, False, y)


def one():
    dense.conv2d_forward_chain(x, w, b, w2, b2, res, True, True, y)


def timeit(fn, cold):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            if cold:
                scr.zero_()
            fn()
    gz = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gz):
        for _ in range(reps):
            if cold:
                scr.zero_()
    def run(gg):
        gg.replay()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gg.replay(); e.record(); e.synchronize()
        return a.elapsed_time(e) * 1e3 / reps
    return run(gr) - (run(gz) if cold else 0.0)


w3 = (torch.randn((64, 1, 1, 256), device="cuda", generator=g) * 0.1).to(torch.bfloat16)
b3 = torch.randn((64,), device="cuda", generator=g)
a1n = torch.empty((N, H, W, 64), device="cuda", dtype=torch.bfloat16)


def three():
    two()
    dense.conv2d_forward(y, w3, b3, None, 1, 0, True, False, a1n)


def one3():
    dense.conv2d_forward_chain(x, w, b, w2, b2, res, True, True, y, w3=w3, bias3=b3, relu3=True, out3=a1n)


for cold in (False, True):
    print("%s: + next conv1: three launches %.1f us, chained %.1f us" % ("cold" if cold else "warm", timeit(three, cold), timeit(one3, cold)))
for cold in (False, True):
    print("%s: two launches %.1f us, chained %.1f us" % ("cold (1 GiB overwritten before each)" if cold else "warm", timeit(two, cold), timeit(one, cold)))
