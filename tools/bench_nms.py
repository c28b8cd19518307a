"""Device time of batched NMS on RPN-like lists (B lists of n boxes).  python tools/bench_nms.py [B] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mxdetection_amd.ops import nms_batched
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(0)
ctr = rng.uniform(0, 1300, (B, n, 2)).astype(np.float32)
wh = np.exp(rng.uniform(np.log(16), np.log(400), (B, n, 2))).astype(np.float32)
boxes = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], 2)).cuda()
counts = torch.full((B,), n, dtype=torch.int32).cuda()
def run():
    return nms_batched(boxes, counts, 0.7)
keep, num = run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        run()
g.replay()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); g.replay(); b.record(); b.synchronize()
print("nms B=%d n=%d: %.1f us per call, kept %s" % (B, n, a.elapsed_time(b) * 100.0, num.cpu().tolist()[:5]))
