"""HIP events that a captured hipGraph records and code OUTSIDE the graph can wait for or time (plumbing: streams and
graphs are torch's / HIP's, nothing here computes).

torch.cuda.Event.record() inside a stream capture only adds a dependency edge: nothing outside the graph can observe it.
An event-record NODE can: during capture the stream's graph and current dependency set are taken with
hipStreamGetCaptureInfo_v2, a node is added with hipGraphAddEventRecordNode and made the stream's new dependency with
hipStreamUpdateCaptureDependencies. Measured on MI355X / ROCm 7.2 (tools/micro/ext_event_probe.py,
tools/micro/ext_event_wait_probe.py): hipEventElapsedTime between two such events gives the device time of the kernels
between them, and hipStreamWaitEvent on another stream waits for the record of the most recent graph launch (the waiting
stream resumed ~10 us after the recording node). hipEventRecordWithFlags(hipEventRecordExternal) itself returns
hipErrorInvalidValue under torch's capture on this runtime, hence the explicit node."""
import ctypes as C

import torch

_hip = None


def _lib():
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")        # resolves to the runtime torch has already mapped (same SONAME)
        h.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        h.hipEventDestroy.argtypes = [C.c_void_p]
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        h.hipEventSynchronize.argtypes = [C.c_void_p]
        h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        h.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        h.hipStreamGetCaptureInfo_v2.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.c_size_t)]
        h.hipGraphAddEventRecordNode.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
        h.hipStreamUpdateCaptureDependencies.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        _hip = h
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class GraphEvent:
    """A HIP event usable from inside a captured graph (record_node) and from plain streams (record / wait)."""

    def __init__(self):
        self.handle = C.c_void_p()
        _check(_lib().hipEventCreate(C.byref(self.handle)), "hipEventCreate")

    def record(self, stream=None):
        """Plain record on a stream that is NOT capturing."""
        s = stream if stream is not None else torch.cuda.current_stream()
        _check(_lib().hipEventRecord(self.handle, C.c_void_p(s.cuda_stream)), "hipEventRecord")

    def record_node(self, stream=None):
        """Record from inside the capture running on `stream` (default: the current stream): an event-record node behind
        everything captured on that stream so far."""
        h = _lib()
        s = C.c_void_p((stream if stream is not None else torch.cuda.current_stream()).cuda_stream)
        status, cid, graph, deps, nd = C.c_int(), C.c_ulonglong(), C.c_void_p(), C.POINTER(C.c_void_p)(), C.c_size_t()
        _check(h.hipStreamGetCaptureInfo_v2(s, C.byref(status), C.byref(cid), C.byref(graph), C.byref(deps), C.byref(nd)),
               "hipStreamGetCaptureInfo_v2")
        if status.value != 1 or not graph:       # hipStreamCaptureStatusActive
            raise RuntimeError("record_node: the stream is not capturing")
        node = C.c_void_p()
        _check(h.hipGraphAddEventRecordNode(C.byref(node), graph, deps, nd.value, self.handle), "hipGraphAddEventRecordNode")
        arr = (C.c_void_p * 1)(node)
        _check(h.hipStreamUpdateCaptureDependencies(s, arr, 1, 1), "hipStreamUpdateCaptureDependencies")   # 1 = set

    def wait(self, stream=None):
        """Make `stream` (not capturing) wait for the most recently launched record of this event."""
        s = stream if stream is not None else torch.cuda.current_stream()
        _check(_lib().hipStreamWaitEvent(C.c_void_p(s.cuda_stream), self.handle, 0), "hipStreamWaitEvent")

    def synchronize(self):
        _check(_lib().hipEventSynchronize(self.handle), "hipEventSynchronize")

    def elapsed_us(self, later):
        ms = C.c_float()
        _check(_lib().hipEventElapsedTime(C.byref(ms), self.handle, later.handle), "hipEventElapsedTime")
        return ms.value * 1e3
