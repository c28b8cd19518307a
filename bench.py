#!/usr/bin/env python
"""bench.py -- images/sec of the Faster R-CNN R50-FPN training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-reduce of the gradient buckets)

A step = forward + RPN/RCNN targets + losses + explicit backward + gradient all-reduce + SGD-momentum update on
a synthetic batch of 2 images of 3x800x1333 (zero-padded to 1344) per GPU, inputs resident in HBM.
Prints ONE JSON line (rank 0). See DESIGN.md section 7 for what each field means.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 2
IM_H, IM_W, PAD_W = 800, 1333, 1344
G_MAX = 100
# algorithmic conv-GEMM work per image: fwd 201.6 GMAC; training with conv1+C2 frozen
TRAIN_CONV_FLOP_PER_IMAGE = 1.142e12
MFMA_PEAK_BF16 = 2.5e15


def synth_batch(rank, step, device):
    """SURVEY.md section 8d synthetic inputs."""
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.zeros((BATCH_PER_GPU, 3, IM_H, PAD_W), dtype=torch.float32)
    img[..., :IM_W] = torch.randn((BATCH_PER_GPU, 3, IM_H, IM_W), generator=g)
    rng = np.random.default_rng(4321 + rank * 1000 + step)
    gt = -np.ones((BATCH_PER_GPU, G_MAX, 5), np.float32)
    for n in range(BATCH_PER_GPU):
        G = int(rng.integers(4, 17))
        for k in range(G):
            w = min(float(np.exp(rng.uniform(np.log(16), np.log(600)))), IM_W - 1)
            h = min(float(np.exp(rng.uniform(np.log(16), np.log(600)))), IM_H - 1)
            x1, y1 = float(rng.uniform(0, IM_W - w)), float(rng.uniform(0, IM_H - h))
            gt[n, k] = [x1, y1, x1 + w - 1, y1 + h - 1, float(rng.integers(1, 81))]
    im_info = torch.tensor([[IM_H, IM_W, 1.0]] * BATCH_PER_GPU, dtype=torch.float32)
    return img.to(device), torch.from_numpy(gt).to(device), im_info.to(device)


class ConvTimer:
    """Roofline measurement for the conv kernel families.

    The timed region replays hipGraphs, so per-launch events cannot sit inside it. Instead one eager step is
    logged (every conv launch with its arguments) and each logged launch is then re-issued REPS times back to back
    inside one hipGraph replayed between two HIP events (GPU saturated, no host launch cost in the measurement). Algorithmic FLOPs
    per launch = 2 * M * Ncols * K. rocprofv3 --kernel-trace --stats of the same command is committed under profiles/."""
    REPS = 8

    def __init__(self):
        self.log = []
        self.logging = False
        self.orig = {}
        self.heaviest = {}    # family -> (flops, seconds) of its largest launch

    def install(self):
        from mxdetection_amd.ops import dense
        timer = self

        def wrap(name, family, flops_of, shape_of):
            fn = getattr(dense, name)
            timer.orig[name] = fn

            def inner(*a, **kw):
                if timer.logging:
                    timer.log.append((family, flops_of(*a, **kw), fn, a, kw, shape_of(*a, **kw)))
                return fn(*a, **kw)
            setattr(dense, name, inner)

        def s_fwd(x, w, *a, **kw):
            return "N=%d %dx%d %d->%d %dx%d" % (x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0], w.shape[1], w.shape[2])

        def s_dgrad(dy, wt, x_shape, KH, KW, *a, **kw):
            return "N=%d %dx%d %d->%d %dx%d" % (x_shape[0], x_shape[1], x_shape[2], x_shape[3], dy.shape[3], KH, KW)

        def s_wgrad(x, dy, KH, KW, *a, **kw):
            return "N=%d %dx%d %d->%d %dx%d" % (x.shape[0], x.shape[1], x.shape[2], x.shape[3], dy.shape[3], KH, KW)

        def f_fwd(x, w, bias=None, residual=None, stride=1, pad=0, relu=False, res_upsample=False, out=None, **_):
            N, H, W, Cin = x.shape
            Cout, KH, KW, _ = w.shape
            Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
            return 2.0 * N * Ho * Wo * Cout * KH * KW * Cin

        def f_dgrad(dy, wt, x_shape, KH, KW, stride=1, pad=0, residual=None, relu_mask=None, accumulate=False, out=None, **_):
            N, Ho, Wo, Cout = dy.shape
            return 2.0 * N * Ho * Wo * Cout * KH * KW * x_shape[3]

        def f_wgrad(x, dy, KH, KW, stride=1, pad=0, dw=None, db=None, accumulate=False, workspace=None, **_):
            N, Ho, Wo, Cout = dy.shape
            return 2.0 * N * Ho * Wo * Cout * KH * KW * x.shape[3]

        def f_chain(x, w, bias, w2, *a, **kw):       # 3x3 (Cin -> 64) + 1x1 (64 -> 256) in one launch
            N, H, W, Cin = x.shape
            w3 = kw.get("w3")          # (+ the next block's conv1, 256 -> 64, when it rides along)
            return 2.0 * N * H * W * (w.shape[0] * 9 * Cin + w2.shape[0] * w.shape[0] +
                                      (w3.shape[0] * w2.shape[0] if w3 is not None else 0))

        def s_chain(x, w, bias, w2, *a, **kw):
            return "N=%d %dx%d %d->%d 3x3 + ->%d 1x1 (chained)" % (x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0],
                                                                   w2.shape[0])

        wrap("conv2d_forward", "conv_igemm_fwd", f_fwd, s_fwd)
        wrap("conv2d_forward_chain", "conv_igemm_fwd", f_chain, s_chain)
        wrap("conv2d_dgrad", "conv_igemm_dgrad", f_dgrad, s_dgrad)
        wrap("conv2d_wgrad", "conv_wgrad", f_wgrad, s_wgrad)
        # grouped weight gradients (one launch pair per parameter bucket) are launches of the same family
        orig_launch = dense.GroupedWgrad.launch
        timer.orig["GroupedWgrad.launch"] = orig_launch

        def grouped_launch(plan, workspace, *rest):
            if timer.logging:
                timer.log.append(("conv_wgrad", plan.flops, orig_launch, (plan, workspace), {}, "group of %d layers" % plan.n))
            return orig_launch(plan, workspace, *rest)
        dense.GroupedWgrad.launch = grouped_launch
        orig_glaunch = dense.GroupedConv.launch
        timer.orig["GroupedConv.launch"] = orig_glaunch

        def grouped_conv_launch(plan):
            if timer.logging:
                timer.log.append(("conv_igemm_fwd" if plan.kind == 0 else "conv_igemm_dgrad", plan.flops, orig_glaunch,
                                  (plan,), {}, "group of %d layers" % plan.n))
            return orig_glaunch(plan)
        dense.GroupedConv.launch = grouped_conv_launch

    @staticmethod
    def time_launch(fn, a, kw, reps):
        """Average device time of one launch: `reps` re-issues captured into a hipGraph (no host launch cost between
        them -- a small conv is shorter than a ctypes call), one warm replay, one replay between HIP events."""
        import torch
        fn(*a, **kw)     # warm (and plans nothing new: shapes are those of the logged step)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn(*a, **kw)
        g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) * 1e-3 / reps

    def measure(self):
        fam = {}
        for family, flops, fn, a, kw, shape in self.log:
            t = self.time_launch(fn, a, kw, self.REPS)
            acc = fam.setdefault(family, [0.0, 0.0, 0])
            acc[0] += flops
            acc[1] += t
            acc[2] += 1
            grouped = fn in (self.orig.get("GroupedWgrad.launch"), self.orig.get("GroupedConv.launch"))   # "heaviest" = the largest SINGLE-layer launch
            if not grouped and (family not in self.heaviest or flops > self.heaviest[family][0]):
                self.heaviest[family] = (flops, t, shape)
        return fam


def cpu_baseline(rank):
    """Bounded CPU leg: the oracle-side torch-CPU/C restatement of the same step on ONE 3x800x1333 image."""
    try:
        from oracle import model_ref
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": "unavailable: %r" % (ex,)}
    return model_ref.timed_cpu_baseline()


def csrc_sha16():
    """Hash of the kernel sources: tracked counter summaries (profiles/*_pmc_step.json) carry it, a stale one yields null."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mxdetection_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def kernel_family(name):
    """conv family of a kernel name of the trace (None: not a conv kernel)."""
    import re
    m = re.search(r"conv_igemm(?:_grouped)?_kernel<([^>]*)>", name)
    if m:
        return "conv_igemm_dgrad" if m.group(1).split(",")[5].strip() == "true" else "conv_igemm_fwd"
    if "wgrad" in name:
        return "conv_wgrad"
    return None


def instep_from_rows(rows):
    """Per-family device time of the last three complete steps of a kernel trace. rows: (start_ns, end_ns, kernel name,
    queue id) of every dispatch. Returns None if the trace holds fewer than four stem launches."""
    rows = sorted(rows)
    # The frozen front end (stem + frozen stages) is a graph of its own that runs on another stream beside the PREVIOUS
    # step's weight-gradient tail (DetectorBase.capture, front_pipeline): its convolution kernels are a family of
    # their own -- the forward launches that follow the stem kernel on the stem's queue, up to that queue's next
    # kernel of another kind. (Their durations are long BECAUSE they share the chip with the tail; the step is shorter.)
    front_left = {}
    tagged = []
    for s0, e0, name, q in rows:
        fam_name = kernel_family(name)
        if "stem_pool_kernel" in name or "stem_conv_kernel" in name:
            front_left[q] = True
        elif front_left.get(q):
            if fam_name == "conv_igemm_fwd":
                fam_name = "conv_frozen_front"
            else:
                front_left[q] = False
        tagged.append((s0, e0, name, fam_name))
    rows = tagged
    stems = [i for i, x in enumerate(rows) if "stem_pool_kernel" in x[2] or "stem_conv_kernel" in x[2]]
    if len(stems) < 4:
        return None
    steps = list(zip(stems[-4:-1], stems[-3:]))       # the last three step periods (stem launch to stem launch)
    fam, wall = {}, 0.0
    share = {}                                        # family -> wall time attributed to it (see below)
    top = {}                                          # kernel symbol -> [device ms, launches]
    for lo, hi in steps:
        win = rows[lo:hi]
        wall += (rows[hi][0] - win[0][0]) * 1e-6
        for s0, e0, name, k in win:
            if k:
                acc = fam.setdefault(k, [0.0, 0])
                acc[0] += (e0 - s0) * 1e-6
                acc[1] += 1
            sym = name.split("(")[0].replace("void ", "")
            t = top.setdefault(sym, [0.0, 0])
            t[0] += (e0 - s0) * 1e-6
            t[1] += 1
        # Kernels of different streams overlap (weight gradients beside the data-gradient chain, the next step's
        # frozen front end beside the weight-gradient tail): the sum of durations then exceeds the wall time they
        # cost. `share`: every instant of the window is split evenly between the kernels running in it, so the shares
        # of all kernels add up to the busy time of the step.
        ev = sorted([(w_[0], 1, i) for i, w_ in enumerate(win)] + [(w_[1], 0, i) for i, w_ in enumerate(win)])
        active, last = set(), None
        for t, kind, i in ev:
            if active and last is not None and t > last:
                dt = (t - last) * 1e-6 / len(active)
                for j in active:
                    k = win[j][3]
                    if k:
                        share[k] = share.get(k, 0.0) + dt
            last = t
            if kind == 1:
                active.add(i)
            else:
                active.discard(i)
    n = float(len(steps))
    dom = max(top.items(), key=lambda kv: kv[1][0])
    return {"families_ms": {k: v[0] / n for k, v in fam.items()}, "kernels": {k: v[1] / n for k, v in fam.items()},
            "families_share_ms": {k: v / n for k, v in share.items()},
            "dominant_kernel": {"name": dom[0], "ms_per_step": dom[1][0] / n, "launches_per_step": dom[1][1] / n},
            "step_ms": wall / n, "steps": len(steps)}


def instep_profile(model, timeout_s=240):
    """Kernel durations INSIDE replayed steps: a child process runs a short replay of the same model under
    `rocprofv3 --kernel-trace` (started before this process touches the GPU, like the ranks of --gpus N); the trace's
    complete steps (a step starts at the stem kernel) give per-family device time as it is in the step: cold operands,
    the other streams' kernels beside it. Returns None when the profiler is not available or fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    d = tempfile.mkdtemp(prefix="mxdet_instep_", dir="/tmp")
    try:
        cmd = [prof, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "tl", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--no-conv-timer", "--no-cpu-baseline", "--no-instep",
               "--model", model]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        files = glob.glob(os.path.join(d, "**", "tl_kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        rows = []
        with open(files[0]) as f:
            for x in csv.DictReader(f):
                rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"], x.get("Queue_Id", "0")))
        return instep_from_rows(rows)
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(n, deadline_s=None, child_cmd=None):
    """Launcher for `python bench.py --gpus N` without torch.distributed.run: one child per GPU with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1. Children are new processes (never a re-exec of one that
    initialised the GPU), each in a session of its own with a parent-death signal, so they cannot outlive the launcher:
    SIGTERM / SIGINT to the launcher, a rank that dies, or the overall deadline (--timeout) stop exactly the child PIDs
    (SIGTERM, then SIGKILL after a grace period). Returns the exit code: 0 only if every rank exited 0."""
    import ctypes
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]

    def pdeathsig():      # child side, before exec: die with the launcher (PR_SET_PDEATHSIG = 1)
        try:
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)
        except Exception:  # noqa: BLE001
            pass
    procs = []
    out0 = tempfile.TemporaryFile()
    cmd = child_cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True, preexec_fn=pdeathsig))
    rcs = [None] * n
    stop = {"why": None}

    def on_signal(signum, _frame):
        stop["why"] = "signal %d" % signum
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}

    def stop_children():
        for r, p in enumerate(procs):
            if rcs[r] is None:
                p.terminate()
        t_end = time.time() + 20
        for r, p in enumerate(procs):
            if rcs[r] is None:
                try:
                    rcs[r] = p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    rcs[r] = p.wait()
    t_start = time.time()
    try:
        # a rank that dies leaves the others waiting in a collective: stop them (exact PIDs) instead of hanging
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if stop["why"] is None and deadline_s and time.time() - t_start > deadline_s:
                stop["why"] = "deadline of %.0f s" % deadline_s
            if stop["why"] is not None or any(rc not in (None, 0) for rc in rcs):
                stop_children()
                break
            time.sleep(0.2)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    if stop["why"] is not None:
        sys.stderr.write("bench.py: launcher stopped its ranks (%s)\n" % stop["why"])
        return 124
    out0.seek(0)
    sys.stdout.write(out0.read().decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed (rank, exit code): %s\n" % bad)
        first = [rc for _, rc in bad if rc > 0]
        return first[0] if first and first[0] < 256 else 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--timeout", type=float, default=0.0,
                    help="--gpus N launcher only: overall deadline in seconds (0 = none); the ranks are stopped when it passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-timer", action="store_true")
    ap.add_argument("--no-instep", action="store_true",
                    help="skip the in-step kernel timing (a child run under rocprofv3 --kernel-trace before the timed run)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-wgrad-stream", action="store_true", help="keep weight gradients on the main stream")
    ap.add_argument("--no-branch-stream", action="store_true", help="keep the RPN training branch on the main stream")
    ap.add_argument("--no-grouped-wgrad", action="store_true", help="two launches per layer instead of per bucket")
    ap.add_argument("--input", default="resident", choices=["resident", "loader"],
                    help="resident = batches already in HBM (the headline contract); loader = every step takes its batch from "
                         "datasets.DetectionLoader: host frames -> pinned -> H2D -> preprocess kernel (PCIe-inclusive rate)")
    ap.add_argument("--model", default="faster_rcnn", choices=["faster_rcnn", "mask_rcnn", "retinanet"],
                    help="faster_rcnn = BASELINE.json headline (configs 1-3); mask_rcnn = config 4; retinanet = config 5 (R101)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes a launcher. It starts N fresh rank processes BEFORE
        # anything here touches the GPU (no torch import yet), relays rank 0's JSON line and fails if any rank fails.
        sys.exit(spawn_ranks(args.gpus, args.timeout or None))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    instep = None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if world == 1 and args.gpus == 1 and not (args.no_instep or args.no_conv_timer or args.no_graph or under_profiler) \
            and args.input == "resident":
        instep = instep_profile(args.model)       # before this process touches the GPU

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()          # does not initialise the GPU
    if ndev < max(1, min(world, local_rank + 1)) or (world > 1 and ndev < world):
        sys.stderr.write("bench.py: --gpus %d needs %d visible devices on this node, found %d\n" % (args.gpus, world, ndev))
        sys.exit(3)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    device = "cuda"
    dist = None
    if world > 1 or os.environ.get("MXDET_FORCE_DIST") == "1":   # the env knob exercises the RCCL path at world 1
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)

    from mxdetection_amd.models import FasterRCNN, RetinaNet
    timer = ConvTimer()
    if not args.no_conv_timer:
        timer.install()
    if args.model == "retinanet":
        model = RetinaNet(device, depth=101, seed=7)
    else:
        model = FasterRCNN(device, depth=50, seed=7, with_mask=(args.model == "mask_rcnn"))
    if not args.no_wgrad_stream:
        model.enable_wgrad_stream()
    if not args.no_branch_stream:
        model.enable_branch_stream()
    if not args.no_grouped_wgrad:
        model.enable_grouped_wgrad()
    if dist is not None:
        model.enable_data_parallel(world)
        model.broadcast_parameters(0)
    # linear-scaling rule from lr 0.02 @ batch 16, at the warm-up start factor 1/3 (random-init weights, no BN)
    lr = 0.02 * (BATCH_PER_GPU * world) / 16.0 / 3.0
    if os.environ.get("MXDET_ABL_LR0") == "1":      # timing-only ablation builds (tools/ab_env.sh): wrong gradients must not feed back
        lr = 0.0
    batches = [synth_batch(rank, s, device) for s in range(4)]
    masks = None
    if args.model == "mask_rcnn":   # filled ellipse inside every GT box, [N,16,H,W] u8 (GT rows 0..15 are the valid ones)
        masks = []
        yy = torch.arange(IM_H, device=device).view(1, 1, IM_H, 1).float()
        xx = torch.arange(PAD_W, device=device).view(1, 1, 1, PAD_W).float()
        for (_, gt, _) in batches:
            b = gt[:, :16]
            cx, cy = 0.5 * (b[..., 0] + b[..., 2]), 0.5 * (b[..., 1] + b[..., 3])
            rx, ry = 0.5 * (b[..., 2] - b[..., 0]) + 0.5, 0.5 * (b[..., 3] - b[..., 1]) + 0.5
            m = (((xx - cx[..., None, None]) / rx[..., None, None]) ** 2 +
                 ((yy - cy[..., None, None]) / ry[..., None, None]) ** 2) <= 1.0
            masks.append((m & (b[..., 4] >= 0)[..., None, None]).to(torch.uint8).contiguous())

    feed = None
    if args.input == "loader":
        # 64 landscape COCO-sized frames, decoded once into host memory (JPEG decode is out of scope: no decoder in the
        # image); every rank walks its own slice of each global batch. The captured step reads bf16 NCHW input.
        from mxdetection_amd.datasets import synthetic_roidb
        from mxdetection_amd.datasets.loader import DetectionLoader
        from mxdetection_amd.datasets.synthetic import synthetic_reader
        roidb = [r for r in synthetic_roidb(160, seed=11) if r["width"] >= r["height"]][:64]
        cache = {r["id"]: synthetic_reader(r) for r in roidb}
        gm = 16 if masks else G_MAX
        loader = DetectionLoader(roidb, BATCH_PER_GPU, device=device, rank=rank, world=world, g_max=gm,
                                 with_masks=bool(masks), reader=lambda e: cache[e["id"]], seed=5, num_workers=4)

        def feed_gen():
            ep = 0
            while True:
                loader.set_epoch(ep)
                for b in loader:
                    gt = b["gt_boxes"]
                    if gm != G_MAX:
                        full = torch.full((BATCH_PER_GPU, G_MAX, 5), -1.0, device=device)
                        full[:, :gm] = gt
                        gt = full
                    yield b["image"], gt, b["im_info"], b.get("gt_masks")
                ep += 1
        feed = feed_gen()
        b0 = next(feed)
        batches = [tuple(t.clone() for t in b0[:3])]
        masks = [b0[3].clone()] if masks else None

    use_graph = not args.no_graph
    if use_graph:
        model.capture(*batches[0], lr=lr, image_offset=rank * BATCH_PER_GPU, gt_masks=masks[0] if masks else None)

    def step(i):
        if feed is not None:
            img, gt, info, mk = next(feed)
        else:
            img, gt, info = batches[i % len(batches)]
            mk = masks[i % len(batches)] if masks else None
        if use_graph:
            return model.replay(img, gt, info, i, gt_masks=mk)
        return model.train_step(img, gt, info, step=i, image_offset=rank * BATCH_PER_GPU, lr=lr, gt_masks=mk)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(args.warmup + i)
    t_host = time.perf_counter() - t0          # host enqueue time of the timed steps (before the device has drained)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_vals = [float(v) for v in torch.cat(list(losses)).cpu().numpy()]
    if feed is not None:
        feed.close()

    # roofline pass (every rank does the same local work; the gradient exchange is switched off for it)
    fam = {}
    if not args.no_conv_timer:
        from mxdetection_amd.models.utils.dp import BucketReducer
        saved = model.reducer
        model.reducer = BucketReducer(model.arena.g, None)
        timer.logging = True
        # launches issued by the frozen front end (stem + frozen stages) are logged as a family of their own
        ff = model.backbone.forward_front

        def front_logged(*a, **kw):
            n0 = len(timer.log)
            out = ff(*a, **kw)
            timer.log[n0:] = [("conv_frozen_front",) + e[1:] if e[0] == "conv_igemm_fwd" else e for e in timer.log[n0:]]
            return out
        model.backbone.forward_front = front_logged
        try:
            model.forward_backward(*batches[0], step=10 ** 6, image_offset=rank * BATCH_PER_GPU,
                                   gt_masks=masks[0] if masks else None)   # eager, logged
        finally:
            model.backbone.forward_front = ff
        timer.logging = False
        torch.cuda.synchronize()
        fam = timer.measure()
        model.reducer = saved

    if rank == 0:
        images = args.steps * BATCH_PER_GPU * world
        value = images / dt
        per_gpu = value / world
        roofline = None
        families = {}
        hbm_frac = None
        if fam:
            # time of a family = its kernels' device time INSIDE replayed steps when the in-step trace exists (time_source
            # says which); the warm figure (each launch re-issued back to back) stays beside it as warm_tflops
            ist = instep["families_ms"] if instep else {}
            for k, (fl, tt, n) in fam.items():
                t_in = ist.get(k, 0.0) * 1e-3
                t_use = t_in if t_in > 0 else tt
                families[k] = {"launches_per_step": n, "avg_ms": round(1e3 * t_use / max(n, 1), 4),
                               "tflops": round(fl / t_use / 1e12, 1), "ms_per_step": round(1e3 * t_use, 3),
                               "share_of_step": round(t_use / (dt / args.steps), 3),
                               "warm_tflops": round(fl / tt / 1e12, 1), "warm_ms_per_step": round(1e3 * tt, 3)}
                if instep and instep["families_share_ms"].get(k):
                    # wall time of the step attributable to the family (overlapping kernels share each instant evenly)
                    sh = instep["families_share_ms"][k] * 1e-3
                    families[k]["attributed_ms_per_step"] = round(1e3 * sh, 3)
                    families[k]["attributed_tflops"] = round(fl / sh / 1e12, 1)
            dom = max(families.items(), key=lambda kv: kv[1]["ms_per_step"])[0]
            fl, tt, n = fam[dom]
            t_use = families[dom]["ms_per_step"] * 1e-3
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(fl / t_use / 1e12, 2), "peak": 2500.0,
                        "unit": "TFLOP/s", "frac": round(fl / t_use / MFMA_PEAK_BF16, 4), "traffic": None,
                        "launches_per_step": n, "avg_launch_ms": round(1e3 * t_use / n, 4),
                        "time_source": ("in-step: rocprofv3 --kernel-trace of %d replayed steps (child process of this run, "
                                        "step %.3f ms under the profiler)" % (instep["steps"], instep["step_ms"])) if instep else
                                       "warm: each conv launch of one step re-issued 8x inside a hipGraph replayed between HIP events",
                        "warm_frac": round(fl / tt / MFMA_PEAK_BF16, 4),
                        "flops_source": "2*M*N*K of every launch of the family logged in one eager step"}
            if instep:
                # durations are per kernel as the trace has them: kernels of different streams overlap (the next step's
                # frozen front end runs beside the weight-gradient tail, weight gradients beside the data-gradient chain),
                # so a family's sum of durations can exceed the wall time it costs; conv_families.*.attributed_* split every
                # instant evenly between the kernels running in it
                roofline["overlap_note"] = "sum of in-step kernel durations; overlapping streams inflate it (see attributed_* in conv_families)"
                dk = instep.get("dominant_kernel")
                if dk:
                    roofline["dominant_kernel_symbol"] = {"name": dk["name"][:96], "ms_per_step": round(dk["ms_per_step"], 3),
                                                          "launches_per_step": round(dk["launches_per_step"], 1)}
                    if "wgrad_mixed_grouped_kernel" in dk["name"]:
                        mixed = sum(e[1] for e in timer.log if e[0] == "conv_wgrad" and len(e[3]) > 0 and
                                    getattr(e[3][0], "grid_big", 0) > 0 and getattr(e[3][0], "grid_wgrad", 0) > 0)
                        if mixed > 0:
                            roofline["dominant_kernel_symbol"]["tflops"] = round(mixed / (dk["ms_per_step"] * 1e-3) / 1e12, 1)
                            roofline["dominant_kernel_symbol"]["frac"] = round(mixed / (dk["ms_per_step"] * 1e-3) / MFMA_PEAK_BF16, 4)
        # `traffic` = HBM-side bytes (L2 misses) of the WHOLE dominant family in one step, from the tracked summary of
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over an eager step (tools/pmc_step.sh; FETCH_SIZE
        # doubled for 16-B-per-lane reads as the microarch guide prescribes). It is measured evidence of this round's
        # kernels, not of this very run: `traffic_source` names the file, and the file carries the hash of the kernel
        # sources it was measured on -- after a kernel change the stale figure is dropped (null), not reported.
        if roofline is not None and args.model == "faster_rcnn":
            src = os.path.join("profiles", "r03_pmc_step.json")
            try:
                with open(os.path.join(ROOT, src)) as f:
                    pm = json.load(f)
                if pm.get("csrc_sha16") != csrc_sha16():
                    raise ValueError("stale counter summary")
                roofline["traffic"] = pm["families"][roofline["kernel"]]["hbm_bytes"]
                roofline["traffic_source"] = src
                roofline["traffic_scope"] = "all %d launches of the family in one step" % pm["families"][roofline["kernel"]]["launches_per_step"]
                total = sum(v.get("hbm_bytes", 0) for v in pm["kernels"].values())
                hbm_frac = round(total / (dt / args.steps) / 8.0e12, 4)    # all kernels of the step / step time / 8 TB/s
            except Exception:  # noqa: BLE001
                roofline["traffic"] = None
        if roofline is not None and timer.heaviest.get(roofline["kernel"]):
            hfl, ht, hshape = timer.heaviest[roofline["kernel"]]
            roofline["heaviest_launch"] = {"shape": hshape, "gflop": round(hfl / 1e9, 1),
                                           "ms": round(1e3 * ht, 4), "achieved": round(hfl / ht / 1e12, 1),
                                           "frac": round(hfl / ht / MFMA_PEAK_BF16, 4)}
        out = {
            "metric": {"faster_rcnn": "images/sec (whole node) Faster R-CNN R50-FPN 3x800x1333",
                       "mask_rcnn": "images/sec (whole node) Mask R-CNN R50-FPN 3x800x1333",
                       "retinanet": "images/sec (whole node) RetinaNet R101-FPN 3x800x1333"}[args.model],
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" if feed is None else "synthetic frames through the input pipeline (host -> H2D -> preprocess)",
            "config": {"workload": {"faster_rcnn": "Faster R-CNN ResNet-50-FPN", "mask_rcnn": "Mask R-CNN ResNet-50-FPN",
                                    "retinanet": "RetinaNet ResNet-101-FPN"}[args.model] +
                       " bf16 train step, batch 2/GPU, 3x800x1333 (padded 1344)",
                       "global_batch": BATCH_PER_GPU * world, "parallelism": "dp%d" % world,
                       "frozen": "stem+C2, frozen BN folded", "optimizer": "SGD momentum 0.9 wd 1e-4",
                       "launch": "hipGraph replay" if use_graph else "eager", "input": args.input,
                       "wgrad_side_stream": not args.no_wgrad_stream,
                       "rpn_branch_stream": not args.no_branch_stream,
                       "grouped_wgrad": not args.no_grouped_wgrad,
                       "filter_prefetch_hints": os.environ.get("MXDET_TUNE_PREFETCH", "1") != "0",
                       "params_trainable": model.num_params()},
            "model_mfma_roofline_frac": (round(per_gpu * TRAIN_CONV_FLOP_PER_IMAGE / MFMA_PEAK_BF16, 4)
                                         if args.model == "faster_rcnn" else None),
            "losses_last_step": loss_vals,
            "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 3),
            "roofline": roofline,
            "hbm_frac": hbm_frac,
            "conv_families": families,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rank)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
