"""Device time of batched NMS on RPN-like lists (B lists of n boxes).  python tools/bench_nms.py [B] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mxdetection_amd.ops import nms_batched
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(0)
# RPN-like: boxes cluster around a few dozen objects (jittered copies), so about half survive at IoU 0.7
nobj = 40
oc = rng.uniform(100, 1200, (B, nobj, 2)).astype(np.float32)
ow = np.exp(rng.uniform(np.log(32), np.log(400), (B, nobj, 2))).astype(np.float32)
pick = rng.integers(0, nobj, (B, n))
ctr = np.take_along_axis(oc, pick[..., None].repeat(2, 2), 1) + rng.normal(0, 12, (B, n, 2)).astype(np.float32)
wh = np.take_along_axis(ow, pick[..., None].repeat(2, 2), 1) * np.exp(rng.normal(0, 0.15, (B, n, 2))).astype(np.float32)
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
