"""Grouped weight-gradient launch of ONE layer, three-tap kernel (3x3 layers) vs one-tap 128x128-tile kernel, device time from a hipGraph
replay between HIP events:  python tools/bench_wgrad.py [target] [minsteps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd import _lib          # noqa: E402
from mxdetection_amd.ops import dense      # noqa: E402

lib = _lib.load()
SHAPES = [  # N, H, W, Cin, Cout, K
    (2, 200, 336, 256, 256, 3), (2, 100, 168, 256, 256, 3), (2, 50, 84, 256, 256, 3), (2, 50, 84, 1024, 256, 1),
    (2, 50, 84, 256, 1024, 1), (2, 25, 42, 512, 512, 3), (2, 25, 42, 2048, 512, 1), (1024, 1, 1, 12544, 1024, 1),
    (2, 200, 336, 256, 256, 1)]


def time_graph(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def main():
    if len(sys.argv) > 1:
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_TARGET"], int(sys.argv[1]))
    if len(sys.argv) > 2:
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_MINSTEPS"], int(sys.argv[2]))
    g = torch.Generator(device="cuda").manual_seed(1)
    sel = os.environ.get("WG_SHAPES")          # e.g. "0,2": only these rows of SHAPES
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    eager = os.environ.get("WG_EAGER") == "1"  # plain launches (PMC passes count per dispatch)
    for N, H, W, Cin, Cout, K in shapes:
        pad = K // 2
        x = torch.randn((N, H, W, Cin), device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn((N, H, W, Cout), device="cuda", generator=g).to(torch.bfloat16)
        dw = torch.empty((Cout, K, K, Cin), device="cuda")
        fl = 2.0 * N * H * W * Cout * K * K * Cin
        out = []
        for big in (1, 0):
            lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_ENABLE"], big)
            plan = dense.GroupedWgrad([(x, dy, K, K, 1, pad, dw, None, False)], "cuda")
            ws = torch.empty((max(plan.workspace_bytes, 256),), dtype=torch.uint8, device="cuda")
            if eager:
                for _ in range(3):
                    plan.launch(ws)
                torch.cuda.synchronize()
                t = 1.0
            else:
                t = time_graph(lambda: plan.launch(ws))
            out.append("%s grid %5d slabs %6.1f MB  %7.1f us %6.1f TF" % ("big  " if big else "small", plan.grid_big or plan.grid_wgrad,
                                                                            plan.workspace_bytes / 1e6, t * 1e6, fl / t / 1e12))
        print("N=%d %dx%d %d->%d %dx%d | %s | %s" % (N, H, W, Cin, Cout, K, K, out[0], out[1]), flush=True)


if __name__ == "__main__":
    main()
