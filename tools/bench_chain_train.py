#!/usr/bin/env python
"""A trainable C3 bottleneck tail (3x3 128->128 + ReLU + mask, 1x1 128->512 + shortcut + ReLU + mask) as two launches vs the
chained launch (mxdet_conv2d_fwd_chain_train): bit comparison of everything backward reads, then graph-timed warm and cold.
  python tools/bench_chain_train.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mxdetection_amd.ops import dense
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, W, CM, CO = 2, 100, 168, 128, 512
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((N, H, W, CM), device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn((CM, 3, 3, CM), device="cuda", generator=g) * 0.04).to(torch.bfloat16)
b = torch.randn((CM,), device="cuda", generator=g)
w2 = (torch.randn((CO, 1, 1, CM), device="cuda", generator=g) * 0.1).to(torch.bfloat16)
b2 = torch.randn((CO,), device="cuda", generator=g)
res = torch.randn((N, H, W, CO), device="cuda", generator=g).to(torch.bfloat16)
outs = [[torch.zeros((N, H, W, CM), device="cuda", dtype=torch.bfloat16), torch.zeros((N, H, W, CM // 8), device="cuda", dtype=torch.uint8),
         torch.zeros((N, H, W, CO), device="cuda", dtype=torch.bfloat16), torch.zeros((N, H, W, CO // 8), device="cuda", dtype=torch.uint8)]
        for _ in range(2)]
scr = torch.empty((1 << 28,), device="cuda", dtype=torch.float32)       # 1 GiB: evicts L2 and the Infinity Cache


def two():
    mid, mb, y, yb = outs[0]
    dense.conv2d_forward(x, w, b, None, 1, 1, True, False, mid, bits_out=mb)
    dense.conv2d_forward(mid, w2, b2, res, 1, 0, True, False, y, bits_out=yb)


def one():
    mid, mb, y, yb = outs[1]
    dense.conv2d_forward_chain_train(x, w, b, w2, b2, res, mid, y, mb, yb)


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


two(); one(); torch.cuda.synchronize()
for k, name in enumerate(("mid", "mid bits", "out", "out bits")):
    a, c = outs[0][k], outs[1][k]
    same = torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, c.view(torch.int16) if c.dtype == torch.bfloat16 else c)
    print("%-9s %s" % (name, "bit-identical" if same else "DIFFERENT (%d elements)" % int((a != c).sum())))
for cold in (False, True):
    print("%s: two launches %.1f us, chained %.1f us" % ("cold" if cold else "warm", timeit(two, cold), timeit(one, cold)))
