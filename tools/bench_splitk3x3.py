"""Verdict item 3: the intake-bound C4 / C5 3x3 layers on wider tiles with cross-workgroup split-K (deterministic fold) against
the heuristic's 64x64 tiles. Device time from a hipGraph replay (warm), forward with bias + ReLU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense
lib = _lib.load()


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


for (N, H, W, C) in ((2, 50, 84, 256), (2, 25, 42, 512), (2, 100, 168, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, C), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((C, 3, 3, C), device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.randn((C,), device="cuda", generator=g)
    y = torch.empty((N, H, W, C), device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * N * H * W * C * 9 * C
    t0 = timeit(lambda: dense.conv2d_forward(x, w, bias, None, 1, 1, True, False, out=y))
    ref = y.clone()
    print("N=%d %dx%d %d->%d 3x3: heuristic %6.1f us %6.1f TF" % (N, H, W, C, C, t0, fl / t0 / 1e6), flush=True)
    ws = torch.empty(8 * y.numel() * 4, dtype=torch.uint8, device="cuda")
    d = dense.conv_desc(N, H, W, C, C, 3, 3, 1, 1, True)
    import ctypes as Ct
    for tile in (0, 1, 2):
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["SPLITK_TILE"], tile)
        for ks in (2, 4, 8):
            if ks > C // 64:
                continue
            def run():
                _lib.check(lib.mxdet_conv2d_fwd_splitk(Ct.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), None, _lib.ptr(y), ks,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "splitk")
            t = timeit(run)
            err = (y.float() - ref.float()).abs().max().item()
            print("   tile %s ksplit %d: %6.1f us %6.1f TF  (max |diff| vs heuristic %.3g)" % (
                ("64x64", "128x128 4 waves", "128x128 8 waves")[tile], ks, t, fl / t / 1e6, err), flush=True)
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["SPLITK_TILE"], -1)
