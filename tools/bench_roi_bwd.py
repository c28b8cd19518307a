"""RoIAlign backward in isolation on the rois of real bench steps (tools/data/bench_rois.npz, from tools/dump_rois.py):
segment form vs table form, graph-timed.  usage: python tools/bench_roi_bwd.py [PH] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mxdetection_amd import _lib
from mxdetection_amd.ops import roi_align_backward_gather
PH = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bench_rois.npz"))
lib = _lib.load()
N, C = 2, 256
shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
scales = [0.25, 0.125, 0.0625, 0.03125]
maps = [torch.zeros((N, H, W, C), dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]
for s in range(6):
    rois = torch.from_numpy(d["rois%d" % s]).cuda()
    levels = torch.from_numpy(d["levels%d" % s]).cuda()
    R = rois.shape[0]
    go = torch.randn((R, PH, PH, C), device="cuda").to(torch.bfloat16)
    res = []
    for form, rows in ((0, 1), (0, 2), (0, 4), (0, 8), (1, 0)):
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["ROI_TABLE"], form)
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["ROI_ROWS"], rows)
        for _ in range(3):
            roi_align_backward_gather(maps, scales, rois, levels, go, 2, 2, accumulate=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            roi_align_backward_gather(maps, scales, rois, levels, go, 2, 2, accumulate=True)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps * 1000.0)
    print("step %d  levels %s  segment form, 1/2/4/8 rows per tile: %.1f %.1f %.1f %.1f us   table form %.1f us" % (
        s, np.bincount(d["levels%d" % s], minlength=6)[2:].tolist(), res[0], res[1], res[2], res[3], res[4]))
