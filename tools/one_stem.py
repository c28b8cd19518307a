"""The fused stem + max-pool launch alone, eager, for rocprofv3: python tools/one_stem.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd.ops import dense      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
img = torch.randn((2, 3, 800, 1344), device="cuda")
w = (torch.randn((64, 7, 7, 3), device="cuda") * 0.1).to(torch.bfloat16)
b = torch.zeros((64,), device="cuda")
out = None
for _ in range(reps):
    out = dense.stem_conv7x7_pool(img, w, b, out)
torch.cuda.synchronize()
