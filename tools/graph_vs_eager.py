"""Debug: loss trajectory of eager vs hipGraph replay on the same batches (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mxdetection_amd.models import FasterRCNN

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0025
batches = [bench.synth_batch(0, s, "cuda") for s in range(4)]
for mode in ("eager", "graph"):
    m = FasterRCNN("cuda", seed=7)
    if mode == "graph":
        m.capture(*batches[0], lr=lr, image_offset=0, warmup=0)
    out = []
    for i in range(14):
        b = batches[i % 4]
        l = m.replay(*b, i) if mode == "graph" else m.train_step(*b, step=i, image_offset=0, lr=lr)
        out.append([round(float(v), 4) for v in torch.cat(l).cpu()])
    print(mode, out)
