"""Data-parallel gradient exchange: bucketed, asynchronous all-reduce of contiguous slices of the flat gradient
arena.

The path shards by image: every rank runs the whole step on its own images; the only exchange
is the gradient sum. Parameters are registered in backward-completion order, so a bucket [lo, hi) is final as soon as
backward has passed the corresponding arena mark and its all-reduce overlaps the rest of backward. Sums are fp32;
the 1/world average is folded into the optimizer's `rescale`.

Transport: on GPUs the collective is the library's own (`mxdet_allreduce_bucket` on an `mxdet_comm_t`, include/mxdet.h:
RCCL over xGMI on the communicator's side stream) -- the same entry a reference-side kvstore replacement would bind
(/root/reference/README.md:37). torch.distributed is then only the control plane: it carries the 128-byte communicator
id to the ranks. With a gloo process group (CPU tests; two ranks sharing one GPU) the exchange itself goes through
torch.distributed.
"""
import ctypes as C
import os


class _Ticket:
    """Handle of one bucket issued through mxdet_allreduce_bucket; wait() orders the CURRENT stream behind its sum."""
    __slots__ = ("comm", "ticket")

    def __init__(self, comm, ticket):
        self.comm, self.ticket = comm, ticket

    def wait(self):
        self.comm.wait(self.ticket)


class RcclComm:
    """mxdet_comm_t of this rank (RCCL communicator + side stream + events behind the C-ABI)."""

    def __init__(self, dist):
        import torch
        from ... import _lib
        self._lib_mod = _lib
        lib = _lib.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(lib.mxdet_comm_unique_id(ident), "comm_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)          # control plane only: 128 bytes
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        torch.cuda.current_stream().synchronize()
        _lib.check(lib.mxdet_comm_create(ident, self.world, self.rank, C.byref(self.handle)), "comm_create")

    def allreduce(self, t):
        """In-place fp32 sum of the contiguous tensor t, ordered behind the current stream, on the side stream."""
        _lib = self._lib_mod
        ticket = C.c_int32()
        _lib.check(_lib.load().mxdet_allreduce_bucket(self.handle, _lib.ptr(t), t.numel(), _lib.stream_ptr(),
                                                      C.byref(ticket)), "allreduce_bucket")
        return _Ticket(self, ticket.value)

    def wait(self, ticket=-1):
        _lib = self._lib_mod
        _lib.check(_lib.load().mxdet_comm_wait(self.handle, ticket, _lib.stream_ptr()), "comm_wait")

    def broadcast(self, t, root=0):
        _lib = self._lib_mod
        _lib.check(_lib.load().mxdet_comm_broadcast(self.handle, _lib.ptr(t), t.numel() * t.element_size(), root,
                                                    _lib.stream_ptr()), "comm_broadcast")

    def close(self):
        if self.handle:
            self._lib_mod.load().mxdet_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def make_comm(dist):
    """The C-ABI communicator when the process group's device backend is RCCL ("nccl"), else None (gloo: the exchange
    stays in torch.distributed). MXDET_DIST_TRANSPORT=torch keeps torch.distributed for the exchange on GPUs too."""
    if dist is None or os.environ.get("MXDET_DIST_TRANSPORT", "capi") == "torch":
        return None
    if "nccl" not in str(dist.get_backend()):
        return None
    return RcclComm(dist)


class BucketReducer:
    def __init__(self, flat_grad, dist=None, max_bucket_elems=8 * 1024 * 1024, comm=None):
        self.g, self.dist, self.cap, self.comm = flat_grad, dist, max_bucket_elems, comm
        self.pending = []
        self.log = []          # (lo, hi) of every all-reduce issued since the last wait()

    def reduce(self, lo, hi):
        """All-reduce g[lo:hi] asynchronously. Returns this bucket's work handles (waiting on them from a stream orders
        that stream after the bucket's sums; wait() does it for all buckets). Through the C-ABI a bucket is ONE message
        (xGMI rings are per-link bound: fewer, larger collectives); through torch.distributed it is split into
        <= cap-element messages."""
        if self.dist is None or hi <= lo:
            return []
        mine = []
        if self.comm is not None:
            mine.append(self.comm.allreduce(self.g[lo:hi]))
            self.log.append((lo, hi))
        else:
            s = lo
            while s < hi:
                e = min(hi, s + self.cap)
                mine.append(self.dist.all_reduce(self.g[s:e], async_op=True))
                self.log.append((s, e))
                s = e
        self.pending.extend(mine)
        return mine

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        log, self.log = self.log, []
        return log
