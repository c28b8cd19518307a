"""Probe: external event-record nodes inside a captured hipGraph (hipEventRecordWithFlags + hipEventRecordExternal), read
back with hipEventElapsedTime after a replay. Prints the per-kernel times of a 3-kernel chain."""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecordWithFlags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]


def ev():
    e = C.c_void_p()
    assert hip.hipEventCreate(C.byref(e)) == 0
    return e


import sys
MODE = sys.argv[1] if len(sys.argv) > 1 else "ext"
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]


hip.hipStreamGetCaptureInfo_v2.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p),
                                           C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.c_size_t)]
hip.hipGraphAddEventRecordNode.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
hip.hipStreamUpdateCaptureDependencies.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]


def rec(e):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if MODE == "node":
        status, cid, graph, deps, nd = C.c_int(), C.c_ulonglong(), C.c_void_p(), C.POINTER(C.c_void_p)(), C.c_size_t()
        rc = hip.hipStreamGetCaptureInfo_v2(st, C.byref(status), C.byref(cid), C.byref(graph), C.byref(deps), C.byref(nd))
        node = C.c_void_p()
        rc2 = hip.hipGraphAddEventRecordNode(C.byref(node), graph, deps, nd.value, e)
        arr = (C.c_void_p * 1)(node)
        rc3 = hip.hipStreamUpdateCaptureDependencies(st, arr, 1, 1)
        print("node rc", rc, rc2, rc3, "deps", nd.value)
        return
    rc = hip.hipEventRecordWithFlags(e, st, 1) if MODE == "ext" else hip.hipEventRecord(e, st)
    print("record rc", rc)


x = torch.randn(4096, 4096, device="cuda")
y = torch.empty_like(x)
evs = [ev() for _ in range(4)]
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
torch.mm(x, x, out=y)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    rec(evs[0])
    torch.mm(x, x, out=y)
    rec(evs[1])
    y.mul_(2.0)
    rec(evs[2])
    torch.mm(y, x, out=y.clone())
    rec(evs[3])
for it in range(3):
    g.replay()
    torch.cuda.synchronize()
    out = []
    for a, b in zip(evs[:-1], evs[1:]):
        ms = C.c_float()
        rc = hip.hipEventElapsedTime(C.byref(ms), a, b)
        out.append((rc, round(ms.value * 1e3, 1)))
    print("replay", it, out)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); e.synchronize()
print("whole replay us", s.elapsed_time(e) * 1e3)
