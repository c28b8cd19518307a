"""Probe: does hipStreamWaitEvent on ANOTHER stream honour an event that is recorded by an event-record NODE inside a
replayed hipGraph (node added through hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode +
hipStreamUpdateCaptureDependencies)?  Graph: E0, matmul A, E1, matmul B, E2 on stream 1; stream 2: wait(E1), record Ec."""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipStreamGetCaptureInfo_v2.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p),
                                           C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.c_size_t)]
hip.hipGraphAddEventRecordNode.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
hip.hipStreamUpdateCaptureDependencies.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]


def ev():
    e = C.c_void_p()
    assert hip.hipEventCreate(C.byref(e)) == 0
    return e


def rec_node(e):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    status, cid, graph, deps, nd = C.c_int(), C.c_ulonglong(), C.c_void_p(), C.POINTER(C.c_void_p)(), C.c_size_t()
    assert hip.hipStreamGetCaptureInfo_v2(st, C.byref(status), C.byref(cid), C.byref(graph), C.byref(deps), C.byref(nd)) == 0
    node = C.c_void_p()
    assert hip.hipGraphAddEventRecordNode(C.byref(node), graph, deps, nd.value, e) == 0
    arr = (C.c_void_p * 1)(node)
    assert hip.hipStreamUpdateCaptureDependencies(st, arr, 1, 1) == 0


x = torch.randn(4096, 4096, device="cuda")
y = torch.empty_like(x)
z = torch.empty_like(x)
E0, E1, E2, Ec = ev(), ev(), ev(), ev()
torch.mm(x, x, out=y)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    with torch.cuda.graph(g, stream=s1):
        rec_node(E0)
        torch.mm(x, x, out=y)
        rec_node(E1)
        torch.mm(x, x, out=z)
        rec_node(E2)
torch.cuda.synchronize()
for it in range(3):
    with torch.cuda.stream(s1):
        g.replay()
    rc = hip.hipStreamWaitEvent(C.c_void_p(s2.cuda_stream), E1, 0)
    rc2 = hip.hipEventRecord(Ec, C.c_void_p(s2.cuda_stream))
    torch.cuda.synchronize()
    out = []
    for a, b in ((E0, E1), (E0, E2), (E0, Ec)):
        ms = C.c_float()
        r = hip.hipEventElapsedTime(C.byref(ms), a, b)
        out.append((r, round(ms.value * 1e3, 1)))
    print("replay", it, "wait rc", rc, rc2, "E0->E1, E0->E2, E0->Ec (us):", out, flush=True)
