"""Which of the step's streams share a hardware queue with the communicator? World size 1 over RCCL, the model set up as bench.py
sets it up, two eager steps (every stream has been used), then: a long kernel on a producer stream, one all-reduce of a scratch
tensor behind it, a tiny kernel on every stream of interest; a stream whose kernel finishes while the long kernel still runs
is clear of the communicator's pending wait.   python tools/micro/exchange_queue_report.py"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from mxdetection_amd.models import FasterRCNN

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
N, H, W = 2, 256, 320
rng = np.random.default_rng(0)
image = torch.randn((N, 3, H, W), generator=torch.Generator().manual_seed(1)).cuda()
gt = -torch.ones((N, 16, 5))
gt[:, 0] = torch.tensor([20.0, 30.0, 120.0, 150.0, 3.0])
info = torch.tensor([[H, W, 1.0]] * N).cuda()
m = FasterRCNN("cuda", seed=7, pre_nms_top_n=1000, post_nms_top_n=1000)
m.enable_wgrad_stream(); m.enable_branch_stream(); m.enable_grouped_wgrad(); m.enable_data_parallel(1)
for st in range(2):
    m.train_step(image, gt.cuda(), info, step=st, lr=0.001)
torch.cuda.synchronize()
comm = m.comm
P = torch.cuda.Stream()
spares = [torch.cuda.Stream() for _ in range(4)]
names = ["default", "weight-gradient (ws.side)", "branch"] + ["spare %d" % i for i in range(4)]
streams = [torch.cuda.default_stream(), m.ws.side, m.branch] + spares
tiny = torch.zeros((64,), device="cuda"); scratch = torch.zeros((64,), device="cuda"); big = torch.empty((1 << 27,), device="cuda")
for s_ in streams + [P]:
    with torch.cuda.stream(s_):
        tiny.add_(0.0)
torch.cuda.synchronize()
done = torch.cuda.Event()
with torch.cuda.stream(P):
    for _ in range(60):
        big.add_(1.0)
    done.record()
    t = comm.allreduce(scratch)
evs = []
for s_ in streams:
    e = torch.cuda.Event()
    with torch.cuda.stream(s_):
        tiny.add_(0.0); e.record()
    evs.append(e)
clear = [False] * len(streams)
t0 = time.perf_counter()
while not done.query() and time.perf_counter() - t0 < 2.0:
    for i, e in enumerate(evs):
        clear[i] = clear[i] or e.query()
t.wait(); torch.cuda.synchronize()
for n_, c in zip(names, clear):
    print("%-28s %s" % (n_, "clear" if c else "HELD UP by the pending all-reduce (or by the producer's queue)"))
dist.destroy_process_group()
