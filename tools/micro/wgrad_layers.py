"""Per-layer weight gradients of the R50-FPN step, one single-layer launch each, in a fixed order (so that a rocprofv3
--pmc / --kernel-trace pass of this script gives per-LAYER HBM-side bytes and durations):
    python tools/micro/wgrad_layers.py list      -> prints the layer table with the algorithmic minimum bytes
    python tools/micro/wgrad_layers.py run       -> runs every layer REPS times (default 1) after one warm-up pass"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# (name, H, W, Cin, Cout, k, stride, count per step); N = 2, H/W = INPUT map
LAYERS = [
    ("c3b0.conv1", 200, 336, 256, 128, 1, 1, 1), ("c3b0.conv2s2", 200, 336, 128, 128, 3, 2, 1),
    ("c3.conv3", 100, 168, 128, 512, 1, 1, 4), ("c3b0.ds", 200, 336, 256, 512, 1, 2, 1),
    ("c3.conv1", 100, 168, 512, 128, 1, 1, 3), ("c3.conv2", 100, 168, 128, 128, 3, 1, 3),
    ("c4b0.conv1", 100, 168, 512, 256, 1, 1, 1), ("c4b0.conv2s2", 100, 168, 256, 256, 3, 2, 1),
    ("c4.conv3", 50, 84, 256, 1024, 1, 1, 6), ("c4b0.ds", 100, 168, 512, 1024, 1, 2, 1),
    ("c4.conv1", 50, 84, 1024, 256, 1, 1, 5), ("c4.conv2", 50, 84, 256, 256, 3, 1, 5),
    ("c5b0.conv1", 50, 84, 1024, 512, 1, 1, 1), ("c5b0.conv2s2", 50, 84, 512, 512, 3, 2, 1),
    ("c5.conv3", 25, 42, 512, 2048, 1, 1, 3), ("c5b0.ds", 50, 84, 1024, 2048, 1, 2, 1),
    ("c5.conv1", 25, 42, 2048, 512, 1, 1, 2), ("c5.conv2", 25, 42, 512, 512, 3, 1, 2),
    ("fpn.lat2", 200, 336, 256, 256, 1, 1, 1), ("fpn.lat3", 100, 168, 512, 256, 1, 1, 1),
    ("fpn.lat4", 50, 84, 1024, 256, 1, 1, 1), ("fpn.lat5", 25, 42, 2048, 256, 1, 1, 1),
    ("fpn.out2/rpn2", 200, 336, 256, 256, 3, 1, 2), ("fpn.out3/rpn3", 100, 168, 256, 256, 3, 1, 2),
    ("fpn.out4/rpn4", 50, 84, 256, 256, 3, 1, 2), ("fpn.out5/rpn5", 25, 42, 256, 256, 3, 1, 2),
    ("fc1", 1, 1024, 12544, 1024, 1, 1, 1), ("fc2", 1, 1024, 1024, 1024, 1, 1, 1),
]


def min_bytes(H, W, Cin, Cout, k, s):
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    xin = 2 * H * W * Cin * 2 if not (k == 1 and s == 2) else 2 * Ho * Wo * Cin * 2
    return xin + 2 * Ho * Wo * Cout * 2 + Cout * k * k * Cin * 4, 2.0 * 2 * Ho * Wo * Cout * k * k * Cin


if sys.argv[1] == "list":
    for i, (n, H, W, Cin, Cout, k, s, c) in enumerate(LAYERS):
        b, f = min_bytes(H, W, Cin, Cout, k, s)
        print("%2d %-14s x%d  min %7.1f MB  %6.2f GFLOP" % (i, n, c, b / 1e6, f / 1e9))
    sys.exit(0)

import torch
from mxdetection_amd.ops import dense
torch.manual_seed(0)
ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
reps = int(os.environ.get("REPS", "1"))
ts = []
for n, H, W, Cin, Cout, k, s, c in LAYERS:
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(2, H, W, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(2, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    dw = torch.empty(Cout, k, k, Cin, device="cuda", dtype=torch.float32)
    ts.append((x, dy, k, s, p, dw))
for rep in range(1 + reps):          # the first pass is the warm-up; the profiler's summary takes the LAST pass
    for (x, dy, k, s, p, dw) in ts:
        dense.conv2d_wgrad(x, dy, k, k, s, p, dw=dw, workspace=ws)
    torch.cuda.synchronize()
print("done", len(ts))
