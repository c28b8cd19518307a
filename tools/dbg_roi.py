import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import synth_boxes
from oracle import oracle
from mxdetection_amd import _lib
from mxdetection_amd.ops import roi_align_backward_gather
def _t(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t
def _bf16_t(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)
rng = np.random.default_rng(19)
N, C = 2, 40
shapes = [(50, 84), (25, 42), (13, 21)]
scales = [0.125, 0.0625, 0.03125]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 160
b = synth_boxes(rng, R, 400, 666)
rois = np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)
levels = rng.integers(3, 6, R).astype(np.int32)
go = oracle.f32_to_bf16_bits(rng.standard_normal((R, 7, 7, C)).astype(np.float32))
feats = [np.zeros((N, H, W, C), dtype=np.uint16) for (H, W) in shapes]
want = oracle.roi_align(feats, scales, rois, levels, 7, 7, 2, 3, grad_out_bits=go)
maps = [torch.full((N, H, W, C), 7.0, dtype=torch.bfloat16, device="cuda") for (H, W) in shapes]
roi_align_backward_gather(maps, scales, _t(rois), _t(levels), _bf16_t(go), 2, 3, accumulate=False)
torch.cuda.synchronize()
for l, (g, w) in enumerate(zip(maps, want)):
    g = g.float().cpu().numpy(); w16 = oracle.round_bf16(w)
    d = np.abs(g - w16)
    bad = d > (2.0 ** -8 * np.abs(w16) + 1e-5 * np.abs(w).max())
    print("level", l, "max diff", d.max(), "bad", bad.sum(), "of", bad.size, "scale", np.abs(w).max())
    if bad.any():
        idx = np.argwhere(bad)
        print(" first bad", idx[:8].tolist())
        for i in idx[:4]:
            print("  got", g[tuple(i)], "want", w16[tuple(i)])
        # per-pixel pattern
        pix = bad.any(axis=-1)
        ys, xs = np.where(pix[0])
        print("  img0 bad pixels:", len(ys), "x%8 hist", np.bincount(xs % 8, minlength=8).tolist())
