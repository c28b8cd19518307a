"""Grouped weight-gradient plans of one bench-like step: workgroups, split-K slab bytes, FLOPs per bucket."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mxdetection_amd.ops import dense
from mxdetection_amd.models import FasterRCNN
orig = dense.GroupedWgrad.__init__
def patched(self, calls, device, fused=False):
    orig(self, calls, device, fused)
    params = sum(c[6].numel() for c in calls)
    print("wgrad group: %3d layers  grid %5d (+%d big)  fold grid %5d  slabs %7.1f MB  params %6.1f MB fp32  %.1f GFLOP  heaviest %s" % (
        len(calls), self.grid_wgrad, self.grid_big, self.grid_reduce, self.workspace_bytes / 1e6, params * 4 / 1e6, self.flops / 1e9, self.heaviest[1]), flush=True)
dense.GroupedWgrad.__init__ = patched
model = FasterRCNN("cuda", depth=50, seed=7)
model.enable_wgrad_stream(); model.enable_branch_stream(); model.enable_grouped_wgrad()
img, gt, info = bench.synth_batch(0, 0, "cuda")
model.train_step(img, gt, info, step=0, lr=0.001)
torch.cuda.synchronize()
