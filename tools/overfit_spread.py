"""Spread of the fixed-batch overfit run (tests/test_gpu_model.py::test_detector_learns_a_fixed_batch) over model seeds and
the weight-gradient kernel choice: final losses after 400 replayed steps + whether predict() recovers the ground truth."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxdetection_amd import _lib
from mxdetection_amd.models import FasterRCNN
lib = _lib.load()
N, H, W = 2, 256, 320
image = torch.randn((N, 3, H, W), generator=torch.Generator().manual_seed(5)).cuda()
gt = -torch.ones((N, 8, 5))
gt[0, 0] = torch.tensor([30.0, 40.0, 150.0, 200.0, 3.0]); gt[0, 1] = torch.tensor([180.0, 60.0, 300.0, 180.0, 17.0])
gt[1, 0] = torch.tensor([60.0, 30.0, 260.0, 230.0, 40.0])
for n in range(N):
    for k in range(8):
        if gt[n, k, 4] > 0:
            x1, y1, x2, y2, c = [int(v) for v in gt[n, k]]
            image[n, :, y1:y2, x1:x2] += torch.tensor([1.5, -1.0, 0.5]).view(3, 1, 1).cuda() * (1 + 0.1 * c)
image *= 0.2
gt = gt.cuda(); info = torch.tensor([[H, W, 1.0]] * N).cuda()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
LR, WARM = float(os.environ.get('LR', '0.01')), float(os.environ.get('WARM', '50'))
for seed in [int(v) for v in os.environ.get('SEEDS', '7,8,9').split(',')]:
    for on in (0, 1):
        lib.mxdet_debug_set_tuning(_lib.TUNING_KEYS["T3_ENABLE"], on)
        m = FasterRCNN("cuda", seed=seed, pre_nms_top_n=1000, post_nms_top_n=300, rois_per_image=128)
        m.enable_wgrad_stream(); m.enable_branch_stream(); m.enable_grouped_wgrad()
        m.capture(image, gt, info, lr=LR)
        first_nan = None
        for it in range(steps):
            losses = m.replay(image, gt, info, it, lr=LR * min(1.0, (it + 1) / WARM))
            if os.environ.get("TRACK_NAN") and first_nan is None and it % 10 == 9:
                v = torch.cat(list(losses)).cpu().numpy()
                if not np.all(np.isfinite(v)):
                    first_nan = (it, v, bool(torch.isfinite(m.arena.w).all()))
        if first_nan:
            print("   first non-finite loss seen at step %d: %s, weights finite: %s" % first_nan)
        torch.cuda.synchronize()
        last = torch.cat(list(losses)).cpu().numpy()
        dets, num = m.predict(image, info, score_thresh=0.3)
        torch.cuda.synchronize()
        print("seed %d three-tap %d: losses %s  detections/img %s (gt 2, 1)" % (seed, on, np.round(last, 4), num.cpu().numpy()), flush=True)
