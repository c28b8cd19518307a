"""Dump the sampled rois / levels of bench-like training steps (gpurun_out/rois.npz) for the RoIAlign micro-benchmark."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mxdetection_amd.models import FasterRCNN
model = FasterRCNN("cuda", depth=50, seed=7)
model.enable_grouped_wgrad()
lr = 0.02 * 2 / 16.0 / 3.0
out = {}
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    img, gt, info = bench.synth_batch(0, s % 4, "cuda")
    model.train_step(img, gt, info, step=s, lr=lr)
    torch.cuda.synchronize()
    ex = model.roi_extractor
    out["rois%d" % s] = ex.rois.float().cpu().numpy()
    out["levels%d" % s] = ex.levels.cpu().numpy()
    lv = out["levels%d" % s]
    r = out["rois%d" % s]
    print("step", s, "levels", np.bincount(lv, minlength=6).tolist(), "mean wh", (r[:, 3] - r[:, 1]).mean(), (r[:, 4] - r[:, 2]).mean())
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/rois.npz", **out)
