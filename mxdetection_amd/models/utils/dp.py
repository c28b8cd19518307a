"""Data-parallel gradient exchange: bucketed, asynchronous all-reduce of contiguous slices of the flat gradient
arena (RCCL over xGMI on the GPU box - backend "nccl" - or gloo in the CPU tests).

The path shards by image: every rank runs the whole step on its own images; the only exchange
is the gradient sum. Parameters are registered in backward-completion order, so a bucket [lo, hi) is final as soon as
backward has passed the corresponding arena mark and its all-reduce overlaps the rest of backward. Sums are fp32;
the 1/world average is folded into the optimizer's `rescale`.
"""


class BucketReducer:
    def __init__(self, flat_grad, dist=None, max_bucket_elems=8 * 1024 * 1024):
        self.g, self.dist, self.cap = flat_grad, dist, max_bucket_elems
        self.pending = []
        self.log = []          # (lo, hi) of every all-reduce issued since the last wait()

    def reduce(self, lo, hi):
        """All-reduce g[lo:hi] asynchronously, split into <= cap-element messages. Returns this bucket's work handles
        (waiting on them from a stream orders that stream after the bucket's sums; wait() does it for all buckets)."""
        if self.dist is None or hi <= lo:
            return []
        s = lo
        mine = []
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
