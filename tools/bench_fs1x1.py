"""Filter-stationary 1x1 kernel vs the general tile, per layer shape of the step (device time from a hipGraph replay, warm)."""
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


SHAPES = [(2, 100, 168, 128, 512), (2, 50, 84, 256, 1024), (2, 25, 42, 512, 2048), (2, 200, 336, 256, 256),
          (2, 100, 168, 512, 256), (2, 100, 168, 128, 256), (2, 50, 84, 256, 512)]
for N, H, W, K, Nc in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, K), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((Nc, 1, 1, K), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn((Nc,), device="cuda", generator=g)
    res = torch.randn((N, H, W, Nc), device="cuda", generator=g).to(torch.bfloat16)
    bits = torch.zeros((N, H, W, Nc // 8), dtype=torch.uint8, device="cuda")
    y = torch.empty_like(res)
    fl = 2.0 * N * H * W * K * Nc
    by = (x.numel() + 2 * res.numel()) * 2 + bits.numel()
    row = []
    for on in (0, 1):
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["FS1X1"], on)
        t1 = timeit(lambda: dense.conv2d_forward(x, w, bias, res, 1, 0, True, False, out=y, bits_out=bits))
        t2 = timeit(lambda: dense.conv2d_dgrad(x, w, (N, H, W, Nc), 1, 1, 1, 0, residual=res, relu_bits=bits, out=y))
        row.append((t1, t2))
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["FS1X1"], -1)
    print("M=%6d K=%3d N=%4d | fwd general %6.1f us, fs %6.1f us (%.2f TB/s) | dgrad general %6.1f, fs %6.1f" % (
        N * H * W, K, Nc, row[0][0], row[1][0], by / row[1][0] / 1e6, row[0][1], row[1][1]), flush=True)
