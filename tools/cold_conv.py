"""How much of a small layer's in-step time is cold operands? One conv launched (a) back to back, (b) after a 1 GiB
buffer was overwritten (weights and activations cold in L2 and in the Infinity Cache), (c) as (b) but with the filter
read once by a tiny kernel just before, (d) as (b) with filter AND input touched. Kernel durations come from
rocprofv3 --kernel-trace (run under tools/prof_cmd.sh); this script only issues the launches in a fixed order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mxdetection_amd.ops import dense
N, H, W, Cin, Cout, K = [int(v) for v in sys.argv[1:7]]
p = K // 2
x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(torch.bfloat16)
y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
big = torch.empty(1 << 28, device="cuda", dtype=torch.float32)     # 1 GiB
def conv(): dense.conv2d_forward(x, w, None, None, 1, p, True, False, y)
for _ in range(3): conv()
torch.cuda.synchronize()
for mode in ("hot", "cold", "filter touched", "filter+input touched"):
    for i in range(8):
        if mode != "hot":
            big.fill_(float(i))
        if mode.startswith("filter"):
            w.float().sum()                 # reads the filter (stays in the Infinity Cache / L2)
        if mode == "filter+input touched":
            x.float().sum()
        conv()
    torch.cuda.synchronize()
print("done")
