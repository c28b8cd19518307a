"""Cost of the NCHW <-> NHWC conversions an NCHW (MXNet) caller pays at the boundary of a plugin slot, on the tensors
of the benchmark step: mxdet_nchw_to_nhwc_bf16 (f32 / bf16 NCHW -> bf16 NHWC) and mxdet_nhwc_to_nchw_f32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mxdetection_amd.ops import dense

def time_graph(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps

tot_in = tot_out = 0.0
for name, (N, C, H, W) in [("C2 / P2-level map", (2, 256, 200, 336)), ("C3", (2, 512, 100, 168)), ("C4", (2, 1024, 50, 84)),
                           ("C5", (2, 2048, 25, 42)), ("pooled rois 1024x256x7x7", (1024, 256, 7, 7))]:
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(N, C, H, W, device="cuda").to(dt)
        t = time_graph(lambda: dense.nchw_to_nhwc(x))
        by = x.numel() * (x.element_size() + 2)
        print("%-26s NCHW %-8s -> NHWC bf16: %7.1f us  %5.2f TB/s" % (name, str(dt).split(".")[1], t * 1e6, by / t / 1e12))
        if dt == torch.float32: tot_in += t
    y = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
    t = time_graph(lambda: dense.nhwc_to_nchw(y))
    print("%-26s NHWC bf16 -> NCHW f32    : %7.1f us  %5.2f TB/s" % (name, t * 1e6, y.numel() * 6 / t / 1e12))
    tot_out += t
print("sum over the five tensors: in (f32) %.0f us, out %.0f us" % (tot_in * 1e6, tot_out * 1e6))
