"""Data-pipeline measurements: the preprocess kernel against the HBM roofline, and the
loader's end-to-end rate (host read -> pinned -> H2D -> kernel) with nothing consuming the batches.

    python tools/bench_loader.py [--images 256] [--masks]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mxdetection_amd.datasets import synthetic_roidb  # noqa: E402
from mxdetection_amd.datasets.loader import DetectionLoader  # noqa: E402
from mxdetection_amd.datasets.synthetic import synthetic_reader  # noqa: E402
from mxdetection_amd.process_data import transform as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--masks", action="store_true")
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    dev = "cuda"
    # --- kernel: 2 COCO-sized frames -> [2,3,800,1344] bf16, timed inside a hipGraph (as bench.py times convs)
    rng = np.random.default_rng(0)
    frames = [torch.from_numpy(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).to(dev) for _ in range(2)]
    pre = T.BatchPreprocessor(pad_to=(800, 1344))
    out = torch.empty((2, 3, 800, 1344), dtype=torch.bfloat16, device=dev)
    pre(frames, [False, True], out=out)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    reps = 20
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                pre(frames, [False, True], out=out)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        g.replay()
        e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    alg = out.numel() * 2 + sum(f.numel() for f in frames)      # bytes written + source bytes read once
    res = {"preprocess_kernel_us": round(us, 2), "algorithmic_MB": round(alg / 1e6, 2),
           "GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000.0, 3)}
    # --- loader: frames cached in host memory (decode is out of scope: no JPEG decoder in the image)
    roidb = [r for r in synthetic_roidb(args.images * 2, seed=1) if r["width"] >= r["height"]][:args.images]
    cache = {r["id"]: synthetic_reader(r) for r in roidb}
    L = DetectionLoader(roidb, 2, with_masks=args.masks, g_max=100 if not args.masks else 16, num_workers=args.workers,
                        reader=lambda e: cache[e["id"]])
    for ep in range(2):
        L.set_epoch(ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for b in L:
            n += b["image"].shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res.update(loader_images=n, loader_images_per_sec=round(n / dt, 1), masks=bool(args.masks), workers=args.workers)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
