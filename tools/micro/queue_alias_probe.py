"""Which streams does a pending wait on ONE stream block? HIP multiplexes streams onto a few hardware queues; a stream that
waits for an event holds up every stream that shares its queue. The probe makes stream W (created last, like the communicator's)
wait for the end of a long kernel on stream L, then launches a tiny kernel on each candidate stream and checks whether it
finishes before the long kernel does.   python tools/micro/queue_alias_probe.py [n_candidates]"""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = "cuda"
x = torch.zeros((1 << 28,), device=dev)           # 1 GiB: a fill takes ~0.3 ms; repeated -> a few ms
tiny = torch.zeros((64,), device=dev)
L = torch.cuda.Stream()
cands = [torch.cuda.Stream() for _ in range(n)]
W = torch.cuda.Stream()                           # "communicator" stream: created after the others
for s in [L, W] + cands:                          # first use of every stream (queue binding happens by now at the latest)
    with torch.cuda.stream(s):
        tiny.add_(0.0)
torch.cuda.synchronize()


def trial(c):
    ev_long = torch.cuda.Event()
    with torch.cuda.stream(L):
        for _ in range(40):
            x.add_(1.0)
        ev_long.record()
    W.wait_event(ev_long)                         # the pending wait
    with torch.cuda.stream(W):
        tiny.add_(0.0)
    ev_c = torch.cuda.Event()
    with torch.cuda.stream(c):
        tiny.add_(0.0)
        ev_c.record()
    t0 = time.perf_counter()
    while not ev_c.query():
        if ev_long.query():
            break
        if time.perf_counter() - t0 > 5.0:
            break
    early = ev_c.query() and not ev_long.query()
    torch.cuda.synchronize()
    return early


print("default stream free of W's wait:", trial(torch.cuda.default_stream()))
for i, c in enumerate(cands):
    print("candidate %2d (stream id %d): %s" % (i, c.stream_id, "free" if trial(c) else "BLOCKED behind W's wait"))
