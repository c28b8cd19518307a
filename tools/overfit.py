"""Overfit one fixed batch (exploration for tests/test_gpu_model.py::test_detector_learns_a_fixed_batch)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd.models import FasterRCNN   # noqa: E402

lr, steps = float(sys.argv[1]), int(sys.argv[2])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
N, H, W = 2, 256, 320
g = torch.Generator().manual_seed(5)
image = torch.randn((N, 3, H, W), generator=g).cuda()
gt = -torch.ones((N, 8, 5))
gt[0, 0] = torch.tensor([30.0, 40.0, 150.0, 200.0, 3.0])
gt[0, 1] = torch.tensor([180.0, 60.0, 300.0, 180.0, 17.0])
gt[1, 0] = torch.tensor([60.0, 30.0, 260.0, 230.0, 40.0])
# paint the boxes into the image so that there is a signal to learn
for n in range(N):
    for k in range(8):
        if gt[n, k, 4] > 0:
            x1, y1, x2, y2, c = [int(v) for v in gt[n, k]]
            image[n, :, y1:y2, x1:x2] += torch.tensor([1.5, -1.0, 0.5]).view(3, 1, 1).cuda() * (1 + 0.1 * c)
image *= scale      # the net is positively homogeneous (ReLU, zero biases): the input scale sets the logit scale
gt = gt.cuda()
info = torch.tensor([[H, W, 1.0]] * N).cuda()
m = FasterRCNN("cuda", seed=7, pre_nms_top_n=1000, post_nms_top_n=300, rois_per_image=128)
m.enable_wgrad_stream(); m.enable_branch_stream(); m.enable_grouped_wgrad()
m.capture(image, gt, info, lr=lr)
for it in range(steps):
    warm = min(1.0, (it + 1) / 50.0)
    losses = m.replay(image, gt, info, it, lr=lr * warm)
    if it % 50 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        print(it, [round(float(v), 4) for v in torch.cat(list(losses)).cpu()], flush=True)
dets, num = m.predict(image, info, score_thresh=0.3)
torch.cuda.synchronize()
d = dets.cpu().numpy(); nn = num.cpu().numpy()
for n in range(N):
    print("image", n, "dets", nn[n])
    for k in range(min(int(nn[n]), 6)):
        print("   ", np.round(d[n, k], 1))
