"""Cost of the 16-bit ReLU-mask operand in the expand (1x1, Cmid -> 4 Cmid) data gradients: the same launch with and
without the mask, warm (graph-timed back to back) and after a 1 GiB overwrite (kernel trace under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mxdetection_amd.ops import dense
big = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
for (N, H, W, Cexp, Cmid) in [(2, 100, 168, 512, 128), (2, 50, 84, 1024, 256), (2, 25, 42, 2048, 512)]:
    dy = torch.randn(N, H, W, Cmid, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cmid, 1, 1, Cexp, device="cuda") * 0.05).to(torch.bfloat16)     # conv1: Cexp -> Cmid
    wt = dense.filter_transpose(w)
    x = torch.randn(N, H, W, Cexp, device="cuda").to(torch.bfloat16)
    res = torch.randn(N, H, W, Cexp, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    for mode in ("mask+res", "res only"):
        m = x if mode == "mask+res" else None
        for i in range(6):
            big.fill_(float(i))
            dense.conv2d_dgrad(dy, wt, tuple(x.shape), 1, 1, 1, 0, residual=res, relu_mask=m, out=out)
        torch.cuda.synchronize()
print("done")
