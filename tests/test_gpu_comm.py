"""The gradient exchange behind the C-ABI (mxdet_comm_t, mxdet_allreduce_bucket): RCCL at world size 1 on the test box.
N > 1 over RCCL needs one device per rank: tests/test_gpu_dist2.py uses it when the box has two devices, the driver's
multi-GPU bench otherwise ("RCCL N>1 unverified" on a one-GPU box)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_allreduce_and_wait(hip):
    import torch
    from mxdetection_amd import _lib
    lib = _lib.load()
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.mxdet_comm_unique_id(ident), "unique_id")
    assert any(ident)                                     # RCCL was found and produced an id
    comm = C.c_void_p()
    _lib.check(lib.mxdet_comm_create(ident, 1, 0, C.byref(comm)), "create")
    try:
        g = torch.randn(5_000_000, device="cuda")
        want = g.clone()
        work = torch.cuda.Stream()
        other = torch.cuda.Stream()
        tickets = []
        with torch.cuda.stream(work):
            g.mul_(2.0)                                   # the bucket becomes final ON THIS STREAM ...
            for lo, hi in ((0, 1_000_000), (1_000_000, 5_000_000)):
                t = C.c_int32(-7)
                _lib.check(lib.mxdet_allreduce_bucket(comm, C.c_void_p(g[lo:hi].data_ptr()), hi - lo,
                                                      C.c_void_p(work.cuda_stream), C.byref(t)), "allreduce")
                tickets.append(t.value)
        assert tickets == [0, 1]
        with torch.cuda.stream(other):                    # ... and is consumed on another one, behind its ticket
            _lib.check(lib.mxdet_comm_wait(comm, tickets[1], C.c_void_p(other.cuda_stream)), "wait")
            got = g.clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want * 2.0)               # world 1: the sum is the identity, ordered behind the mul
        _lib.check(lib.mxdet_comm_wait(comm, -1, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "wait all")
        _lib.check(lib.mxdet_comm_broadcast(comm, C.c_void_p(g.data_ptr()), g.numel() * 4, 0,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "broadcast")
        torch.cuda.synchronize()
        assert torch.equal(g, want * 2.0)
        # more buckets than event slots in flight: older tickets are covered by the slot's younger owner
        for i in range(70):
            _lib.check(lib.mxdet_allreduce_bucket(comm, C.c_void_p(g.data_ptr()), 1024,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream), None), "allreduce")
        _lib.check(lib.mxdet_comm_wait(comm, 3, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "wait old")
        torch.cuda.synchronize()
        assert lib.mxdet_allreduce_bucket(comm, None, 4, None, None) == -1
    finally:
        assert lib.mxdet_comm_destroy(comm) == 0
