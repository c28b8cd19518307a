#!/usr/bin/env python
"""Run one conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace on a single kernel).
  python tools/bench_one_conv.py fwd|dgrad|wgrad N H W Cin Cout K stride [reps] [cfg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense

kind = sys.argv[1]
N, H, W, Cin, Cout, K, s = [int(v) for v in sys.argv[2:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
cfg = int(sys.argv[10]) if len(sys.argv) > 10 else 0
_lib.load().mxdet_debug_force_conv_cfg(cfg)
if kind == "wgrad":
    _lib.load().mxdet_debug_force_wgrad_ksplit(cfg)
p = K // 2
x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
dy = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
wt = dense.filter_transpose(w)
y = torch.empty(N, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
dx = torch.empty_like(x)
dw = torch.empty(Cout, K, K, Cin, device="cuda")
ws = torch.empty(dense.conv2d_wgrad_workspace_bytes(x.shape, Cout, K, K, s, p), dtype=torch.uint8, device="cuda")
def run():
    if kind == "fwd":
        dense.conv2d_forward(x, w, None, None, s, p, True, False, y)
    elif kind == "dgrad":
        dense.conv2d_dgrad(dy, wt, tuple(x.shape), K, K, s, p, out=dx)
    else:
        dense.conv2d_wgrad(x, dy, K, K, s, p, dw=dw, workspace=ws)
for _ in range(3):
    run()
torch.cuda.synchronize()
if os.environ.get("MXDET_EAGER_TIMING"):   # counter collection wants plain dispatches, not graph replays
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    print("%s eager x%d done" % (kind, reps))
    sys.exit(0)
# device time only: reps launches inside one hipGraph (a small conv is shorter than a ctypes call)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(reps):
        run()
g.replay()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
g.replay()
b.record(); b.synchronize()
t = a.elapsed_time(b) * 1e-3 / reps
fl = 2.0 * N * Ho * Wo * Cout * K * K * Cin
print("%s N%d %dx%d Cin%d Cout%d k%d s%d cfg%d: %.1f us  %.1f TFLOP/s" % (kind, N, H, W, Cin, Cout, K, s, cfg, t * 1e6, fl / t / 1e12))

lib = _lib.load()
if hasattr(lib, "mxdet_debug_read_conv_stamps"):
    import ctypes as C
    buf = (C.c_uint64 * 16)()
    lib.mxdet_debug_read_conv_stamps(buf)
    for b in range(2):
        v = list(buf[b * 8:b * 8 + 6])
        names = ["geometry", "first stage landed", "K loop", "drain+barrier", "epilogue"]
        print("  block %s: " % ("0" if b == 0 else "300") + "  ".join("%s %d" % (n, v[i + 1] - v[i]) for i, n in enumerate(names)) + "  total %d cycles" % (v[5] - v[0]))
