"""The box head's first fully connected layer (1,024 rois x 12,544 -> 1,024) forward (split-K, tile choices) and data
gradient (forced tile configurations), device time from a hipGraph replay."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd import _lib
from mxdetection_amd.ops import dense
lib = _lib.load()


def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


R, K, N = 1024, 12544, 1024
x = torch.randn(R, 1, 1, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, 1, 1, K, device="cuda") * 0.01).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
y = torch.empty(R, 1, 1, N, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(64 * R * N * 4, dtype=torch.uint8, device="cuda")
fl = 2.0 * R * K * N
for tile in (0, 1, 2):
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["SPLITK_TILE"], tile)
    for ks in (1, 2, 4, 7, 8, 14, 16):
        try:
            t = timeit(lambda: dense.conv2d_forward_splitk(x, w, bias, None, True, ks, y, ws))
            print("fwd tile %d ksplit %2d: %6.1f us %6.1f TF" % (tile, ks, t, fl / t / 1e6), flush=True)
        except Exception as e:  # noqa: BLE001
            print("fwd tile %d ksplit %d: %s" % (tile, ks, str(e)[:80]))
lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["SPLITK_TILE"], -1)
dy = torch.randn(R, 1, 1, N, device="cuda").to(torch.bfloat16)
wt = dense.filter_transpose(w)
dx = torch.empty_like(x)
for cfg in (0, 40, 41, 45, 46, 49, 15):
    lib.mxdet_debug_force_conv_cfg(cfg)
    t = timeit(lambda: dense.conv2d_dgrad(dy, wt, (R, 1, 1, K), 1, 1, 1, 0, out=dx))
    print("dgrad cfg %2d: %6.1f us %6.1f TF" % (cfg, t, fl / t / 1e6), flush=True)
lib.mxdet_debug_force_conv_cfg(0)
