"""One grouped weight-gradient launch (a single layer), eager, for rocprofv3: python tools/one_wgrad.py big|small N H W Cin Cout K [target] [minsteps] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd import _lib          # noqa: E402
from mxdetection_amd.ops import dense      # noqa: E402

lib = _lib.load()
big = sys.argv[1] == "big"
N, H, W, Cin, Cout, K = [int(v) for v in sys.argv[2:8]]
if len(sys.argv) > 8:
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_TARGET"], int(sys.argv[8]))
if len(sys.argv) > 9:
    lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_MINSTEPS"], int(sys.argv[9]))
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 5
lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_ENABLE"], int(big))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((N, H, W, Cin), device="cuda", generator=g).to(torch.bfloat16)
dy = torch.randn((N, H, W, Cout), device="cuda", generator=g).to(torch.bfloat16)
dw = torch.empty((Cout, K, K, Cin), device="cuda")
plan = dense.GroupedWgrad([(x, dy, K, K, 1, K // 2, dw, None, False)], "cuda")
ws = torch.empty((max(plan.workspace_bytes, 256),), dtype=torch.uint8, device="cuda")
print("grid big %d small %d reduce %d slabs %.1f MB" % (plan.grid_big, plan.grid_wgrad, plan.grid_reduce, plan.workspace_bytes / 1e6))
for _ in range(reps):
    plan.launch(ws)
torch.cuda.synchronize()
