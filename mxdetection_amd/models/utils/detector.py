"""Shared training-step machinery of the detectors: parameter arena bookkeeping, bucketed gradient exchange, hipGraph
capture / replay of the whole step, side-stream weight gradients, optimizer step."""
import contextlib
import os
import warnings

import torch

from .dp import BucketReducer, make_comm
from .layers import ParamArena, Workspace


# TIMING-ONLY ablations (tools/ablate_step.sh): MXDET_ABL_SKIP=front,sgd,transpose,wgrad leaves the named component out of the
# replayed step -- results are wrong, only the step time means anything (the marginal cost of a component in the overlapped
# schedule, which the sum of its kernel durations overstates). Never set outside that tool.
_PROBE_STREAMS = []          # candidate streams of DetectorBase._stream_clear_of_the_exchange
_ABL = frozenset(t for t in os.environ.get("MXDET_ABL_SKIP", "").split(",") if t)

class DetectorBase:
    def _init_base(self, device):
        self.device = device
        self.arena = ParamArena(device)
        self.ws = Workspace(device)
        self.planned = None
        self.dist = None
        self.comm = None
        self.world = 1
        self.segments = None
        self._cap = False
        self._cur_graph = None
        self._tr_table = None
        self.static_extra = {}
        # single GPU: no exchange to overlap, so everything behind the heads is ONE bucket (see _bucket_here)
        self.bucket_merge = os.environ.get("MXDET_TUNE_BUCKETS", "123")
        self.branch = None
        self._upd = None          # (lr, momentum, wd) while a training step wants its buckets updated as they finish
        self._upd_done = []       # arena ranges already updated in this step
        self._tr_ranges = {}
        self._buckets = []        # (lo, hi) of every bucket exchanged in the captured step
        self._seen_buckets = set()
        self._final_join_opt = False
        self._cap_opt = None      # (lr, momentum, wd) while capturing per-bucket update graphs (N > 1)
        self.opt_stream = None
        # Captured step, frozen front end as a graph of its own (capture(): front_pipeline): the stem + frozen stages of
        # batch k run on the front stream while step k-1 is still in its weight-gradient tail (they depend on the image
        # only). _tail_event is recorded by a node of the main graph where that tail begins.
        self.front_pipeline = os.environ.get("MXDET_TUNE_FRONT_PIPE", "1") != "0"
        self._seg_markers = []    # capture: bucket markers of the open main segment (exchange schedule, see _reduce)
        self._deferred = []       # capture: per-bucket graphs still to be captured (_capture_deferred)
        self._front = None        # {"graphs": [g0, g1], "segments": [s0, s1], "losses": [l0, l1], "stream", "ready", "count"}
        self._tail_event = None

    def _finalize_params(self, layers, frozen_layers=()):
        self.layers = layers
        self.frozen_layers = list(frozen_layers)
        self.arena.finalize()
        for l in self.layers:
            l.materialize()
        self.arena.refresh_bf16()
        self.refresh_transposed()
        self.reducer = BucketReducer(self.arena.g, None)

    def export_params(self):
        """name -> fp32 CPU tensor of every parameter as the kernels see it (bf16 filters, fp32 biases)."""
        out = {"stem.weight": self.backbone.stem_w.float().cpu(), "stem.bias": self.backbone.stem_b.float().cpu()}
        frozen = [l for st in self.backbone.stages for b in st for l in b.layers() if not l.trainable]
        for l in self.layers + frozen:
            out[l.name + ".weight"] = l.w_bf16.float().cpu()
            if l.has_bias:
                out[l.name + ".bias"] = l.bias_f32.float().cpu()
        return out

    # ---- checkpoints: MXNet NDArray-list container, MXNet tensor layouts ----

    def _named_tensors(self):
        """(name, kind, tensor, layer) of every stored parameter: trainable master weights (fp32 arena views), frozen
        filters (bf16) and folded frozen-BN shifts (fp32). layer is the owning ConvLayer (None for the stem)."""
        out = [("stem.weight", "frozen", self.backbone.stem_w, None), ("stem.bias", "frozen", self.backbone.stem_b, None)]
        frozen = [l for st in self.backbone.stages for b in st for l in b.layers() if not l.trainable]
        seen = set()
        for l in list(self.layers) + frozen:
            if id(l) in seen:
                continue
            seen.add(id(l))
            if l.trainable:
                out.append((l.name + ".weight", "w", self.arena.view(l.wi, "w"), l))
                if l.train_bias:
                    out.append((l.name + ".bias", "w", self.arena.view(l.bi, "w"), l))
                elif l.has_bias:
                    out.append((l.name + ".bias", "frozen", l.frozen_bias, l))
            else:
                out.append((l.name + ".weight", "frozen", l.w_bf16, l))
                if l.has_bias:
                    out.append((l.name + ".bias", "frozen", l.bias_f32, l))
        return out

    @staticmethod
    def _to_mx(t, layer):
        """This repo's tensor -> the array MXNet stores for the same parameter: alignment-padding output channels
        (rows past cout_real: fused / padded head outputs) are dropped; convolution filters [O,KH,KW,I] -> OIHW;
        fully connected layers (`fc_in_hwc` set: a 1x1 'convolution' over flattened features) -> 2-D [O, I], with the
        input axis reordered from this repo's (H, W, C) flatten to MXNet's (C, H, W) when the input was spatial."""
        real = layer.cout_real if layer is not None else t.shape[0]
        t = t[:real].float()
        if t.dim() == 4:
            hwc = getattr(layer, "fc_in_hwc", None) if layer is not None else None
            if hwc is not None:
                O = t.shape[0]
                if len(hwc) == 3:
                    t = t.reshape(O, *hwc).permute(0, 3, 1, 2)
                t = t.reshape(O, -1)
            else:
                t = t.permute(0, 3, 1, 2)
        return t.contiguous().cpu().numpy()

    @staticmethod
    def _from_mx(a, like, layer):
        """Inverse of _to_mx onto a tensor shaped like `like` (padding channels zero)."""
        src = torch.from_numpy(a).to(like.device)
        if like.dim() == 4:
            hwc = getattr(layer, "fc_in_hwc", None) if layer is not None else None
            if hwc is not None:
                O = src.shape[0]
                if len(hwc) == 3:
                    src = src.reshape(O, hwc[2], hwc[0], hwc[1]).permute(0, 2, 3, 1)
                src = src.reshape(O, 1, 1, -1)
            else:
                src = src.permute(0, 2, 3, 1)
        out = torch.zeros(like.shape, dtype=torch.float32, device=like.device)
        assert tuple(src.shape[1:]) == tuple(like.shape[1:]) and src.shape[0] <= like.shape[0], \
            "checkpoint %s vs model %s" % (tuple(src.shape), tuple(like.shape))
        out[:src.shape[0]] = src
        return out

    def save_checkpoint(self, path):
        """Write every parameter ("arg:<name>", fp32, in the layout MXNet keeps it in: convolution filters OIHW, fully
        connected weights 2-D [out, C*H*W], alignment padding stripped -- see _to_mx) and the SGD momentum of the
        trainable ones ("aux:momentum:<name>") as an MXNet 1.3.0 `.params` file (utils/params_io.py). The byte layout
        of the container is this repo's reading of MXNet's NDArray::Save; no MXNet-written file exists here to pin it."""
        from ...utils import save_params
        blob = {}
        by_name = {}
        for name, kind, t, layer in self._named_tensors():
            blob["arg:" + name] = self._to_mx(t, layer)
            by_name[name] = layer
        for i, e in enumerate(self.arena.entries):
            blob["aux:momentum:" + e[0]] = self._to_mx(self.arena.view(i, "m"), by_name.get(e[0]))
        save_params(path, blob)

    def load_checkpoint(self, path, strict=True):
        """Inverse of save_checkpoint; refreshes the bf16 / transposed working copies. Returns the names not found."""
        from ...utils import load_params
        blob = load_params(path)
        missing = []
        by_name = {}
        for name, kind, t, layer in self._named_tensors():
            by_name[name] = layer
            a = blob.get("arg:" + name)
            if a is None:
                missing.append(name)
                continue
            t.copy_(self._from_mx(a, t, layer).to(t.dtype))
        for i, e in enumerate(self.arena.entries):
            a = blob.get("aux:momentum:" + e[0])
            if a is not None:
                m = self.arena.view(i, "m")
                m.copy_(self._from_mx(a, m, by_name.get(e[0])))
        if strict and missing:
            raise KeyError("checkpoint lacks %d parameters, e.g. %s" % (len(missing), missing[:3]))
        self.arena.refresh_bf16()
        self.refresh_transposed()
        return missing

    def export_grads(self):
        """name -> fp32 CPU gradient of every trainable parameter."""
        return {e[0]: self.arena.view(i, "g").float().cpu() for i, e in enumerate(self.arena.entries)}

    def num_params(self):
        return sum(e[3] for e in self.arena.entries)

    def refresh_transposed(self):
        """[Cout,KH,KW,Cin] -> [Cin,KH,KW,Cout] copies for dgrad: one batched launch for all trainable filters."""
        from ...ops import dense
        if getattr(self, "_tr_table", None) is None:
            pairs = [(l.w_bf16, l.wt) for l in self.layers if l.trainable]
            self._tr_table = dense.make_transpose_table(pairs, self.device)
        dense.filter_transpose_batched(*self._tr_table)

    def enable_wgrad_stream(self):
        """Issue weight-gradient kernels on a second stream (overlaps them with the data-gradient chain)."""
        self.ws.side = torch.cuda.Stream()

    def enable_branch_stream(self):
        """Run the RPN training branch on its own stream, concurrently with the proposal / RoI-head chain."""
        self.branch = torch.cuda.Stream()

    @contextlib.contextmanager
    def _branch_ctx(self):
        """Run the body on the branch stream, concurrently with what the caller issues next on the main stream, until
        _join_branch(). Under capture the body becomes a hipGraph of its own, replayed on the branch stream: two chains
        forked INSIDE one hipGraph did not overlap on this runtime (the second chain's first kernel started about a
        millisecond after the fork, whatever the capture order -- profiles/r01_f_fork_inside_graph.txt), while separate
        graph launches on two streams are at least plain stream semantics. (Measured afterwards: the second chain still
        starts late whenever the first one is running kernels with more workgroups than the chip holds -- the
        dispatcher drains a grid before it serves another queue -- so the gain over the in-graph fork is small; what
        the split did uncover is that memset NODES at the root of a graph do not order against the kernels behind
        them, hence zero_async() in csrc/common.h.)"""
        if self.branch is None:
            yield
            return
        cur = torch.cuda.current_stream()
        if not self._cap:
            self.branch.wait_stream(cur)
            with torch.cuda.stream(self.branch):
                yield
            return
        self.ws.join()
        self._seg_end()
        self.segments.append(("fork",))
        g = torch.cuda.CUDAGraph()
        self.branch.wait_stream(cur)
        with torch.cuda.stream(self.branch):
            # own memory pool: this graph runs concurrently with the main segments, so temporaries allocated while
            # capturing it must not share (time-multiplexed) memory with theirs
            g.capture_begin(pool=self._pool_branch, capture_error_mode="thread_local")
            yield
            g.capture_end()
        cur.wait_stream(self.branch)
        self.segments.append(("branch", g))
        self._seg_begin()

    def _join_branch(self):
        if self.branch is None:
            return
        if not self._cap:
            torch.cuda.current_stream().wait_stream(self.branch)
            return
        self.ws.join()            # a segment cannot end with weight-gradient work still forked
        self._seg_end()
        self.segments.append(("join",))
        self._seg_begin()

    def enable_data_parallel(self, world_size):
        import torch.distributed as dist
        self.dist = dist
        self.world = world_size
        # RCCL process group: the exchange goes through the library's own collective (mxdet_allreduce_bucket);
        # gloo (CPU tests, two ranks on one GPU): through torch.distributed
        self.comm = make_comm(dist)
        self.reducer = BucketReducer(self.arena.g, dist, comm=self.comm)
        # with a gradient exchange the buckets stay fine (five: every all-reduce but the last overlaps the rest of
        # backward); their weight gradients run as side-stream graphs (_reduce), so fine buckets cost nothing here
        self.bucket_merge = os.environ.get("MXDET_TUNE_BUCKETS", "")

    def broadcast_parameters(self, root=0):
        """Replicate rank `root`'s master weights (and refresh the bf16 / transposed working copies)."""
        if getattr(self, "comm", None) is not None:
            self.comm.broadcast(self.arena.w, root)
        else:
            self.dist.broadcast(self.arena.w, root)
        self.arena.refresh_bf16()
        self.refresh_transposed()

    def _bucket_here(self, point):
        """Reduce points of backward, in order: 0 heads, 1 FPN, 2 layer4, 3 layer3 (layer2 always closes the last
        bucket). A point named in bucket_merge does not close its bucket: the parameters join the next one. Fewer,
        larger buckets mean fewer and better-filled grouped weight-gradient / fold / update launches and fewer
        interruptions of the dgrad chain (whole-step A/B: +1.3…2 % for two buckets instead of five at N = 1)."""
        return str(point) not in self.bucket_merge

    def _guard_replan(self, key):
        """plan() at a new input shape reallocates buffers a captured step holds by address."""
        if self.segments is not None and self.planned is not None and self.planned != key:
            raise RuntimeError("the captured training step holds the buffers planned for %s; a call at %s would free "
                               "them (build a second model for another input shape)" % (self.planned, key))

    def enable_grouped_wgrad(self):
        """Issue the weight gradients of each bucket (box/mask heads, FPN, every ResNet stage) as one grouped launch."""
        self.ws.grouping = True
        if getattr(self, "ws_rpn", None) is not None:
            self.ws_rpn.grouping = True

    def _reduce(self, lo, hi, pre=None):
        """Close the gradient bucket [lo, hi). pre: another workspace whose recorded weight gradients belong to this
        bucket and go out first, on the same stream (the RPN head's, when they were not issued inside the branch)."""
        side_graph = (self._cap and self.dist is not None and hi > lo and self.ws.side is not None and self.ws.grouping
                      and bool(self.ws.pending) and os.environ.get("MXDET_TUNE_WGRAD_GRAPH", "1") == "1")
        if not side_graph:
            if pre is not None:
                pre.side = self.ws.side
                pre.flush()
            self.ws.flush()           # grouped mode: the bucket's recorded weight gradients go out now
            if self.dist is not None:
                # the all-reduce is ordered on the main stream: wait for the side stream's weight gradients. Without an
                # exchange the bucket's update follows its weight gradients ON the side stream and the main stream never
                # waits (optimizer_step / segment ends join the side stream).
                self.ws.join()
        if self._cap:
            if self.dist is not None and hi > lo and side_graph:
                # No cut of the main graph: an event-record NODE marks the point where this bucket's dy / x exist, and the
                # bucket's weight gradients (a graph of their own, replayed on the side stream behind that event), its
                # all-reduce (issued from the side stream, behind them) and its update (a graph on the optimizer stream,
                # behind the all-reduce's ticket) never touch the main stream. The two small graphs are captured after
                # the main capture (_capture_deferred): the recorded weight-gradient calls are stashed here. (Before, the
                # main stream was cut into a segment per bucket and every cut was a 14-32 us hole: -2.9 % at world 1.)
                from ...utils.hipgraph import GraphEvent
                k = len(self._buckets)
                self._buckets.append((lo, hi))
                ev = GraphEvent()
                ev.record_node()
                mw = ["wgrad", None, ev]
                items_pre = []
                if pre is not None:
                    items_pre, pre.pending = pre.pending, []
                items, self.ws.pending = self.ws.pending, []
                self._deferred.append(("wgrad", mw, pre, items_pre, items))
                self._seg_markers.append(mw)
                self._seg_markers.append(("reduce", lo, hi, k, True))
                if self._cap_opt is not None:
                    mu = ["update", None, k]
                    self._deferred.append(("update", mu, lo, hi))
                    self._seg_markers.append(mu)
            elif self.dist is not None and hi > lo:   # no side stream: cut the graph here, the all-reduce runs between segments
                self._seg_end()
                k = len(self._buckets)
                self._buckets.append((lo, hi))
                self.segments.append(("reduce", lo, hi, k, False))
                if self._cap_opt is not None:
                    # The bucket's update is a small graph of its own, replayed on the optimizer stream once that
                    # stream has waited for the bucket's all-reduce: it overlaps the rest of backward exactly like
                    # the single-GPU path, and the main stream never waits for a collective before the end of the step.
                    g = torch.cuda.CUDAGraph()
                    cur = torch.cuda.current_stream()
                    self.opt_stream.wait_stream(cur)
                    with torch.cuda.stream(self.opt_stream):
                        g.capture_begin(pool=self._pool_opt, capture_error_mode="thread_local")
                        self._apply_update(lo, hi, self._cap_opt, 1.0 / self.world)
                        g.capture_end()
                    cur.wait_stream(self.opt_stream)
                    self.segments.append(("update", g, k))
                self._seg_begin()
        else:
            self.reducer.reduce(lo, hi)
            if hi > lo:
                self._seen_buckets.add((lo, hi))
        if self._upd is not None and self.dist is None and hi > lo:
            self._update_range(lo, hi)

    def _update_range(self, lo, hi):
        """SGD-momentum update + bf16 / transposed working copies of one finished bucket, on the weight-gradient stream:
        nothing issued so far in this step reads these parameters any more, so the update overlaps the rest of the
        backward pass instead of forming a serial tail after it (single-GPU path; with a gradient exchange the update
        follows the last all-reduce, see optimizer_step)."""
        ctx = self.ws.fork()
        with (ctx if ctx is not None else contextlib.nullcontext()):
            self._apply_update(lo, hi, self._upd, 1.0)
        self._upd_done.append((lo, hi))

    def _transpose_table(self, lo, hi):
        from ...ops import dense
        key = (lo, hi)
        if key not in self._tr_ranges:
            a = self.arena
            pairs = [(l.w_bf16, l.wt) for l in self.layers if l.trainable and lo <= a.offset_of(l.wi) < hi]
            self._tr_ranges[key] = dense.make_transpose_table(pairs, self.device) if pairs else None
        return self._tr_ranges[key]

    def _apply_update(self, lo, hi, hyper, rescale):
        """SGD-momentum on arena[lo:hi] + refresh of the bf16 / transposed working copies of that range."""
        from ...ops import dense
        lr, momentum, wd = hyper
        a = self.arena
        if "sgd" not in _ABL:
            dense.sgd_momentum_update(a.w[lo:hi], a.g[lo:hi], a.m[lo:hi], a.wb[lo:hi], lr, momentum, wd, rescale)
        table = self._transpose_table(lo, hi)
        if table is not None and "transpose" not in _ABL:
            dense.filter_transpose_batched(*table)

    # ---- hipGraph capture of the whole step (static shapes): removes ~450 host launches per step ----

    def _seg_begin(self):
        self._cur_graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (RCCL's watchdog, loader workers) must not invalidate the capture
        self._cur_graph.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _seg_end(self):
        """Close the current segment. A segment in which nothing was launched (the step opens with a fork to the branch
        stream) is dropped instead of being replayed as an empty graph every step; torch reports that case with a
        warning at capture_end, which is the only place the node count is visible from Python."""
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            self._cur_graph.capture_end()
        empty = False
        for w in rec:
            if "Graph is empty" in str(w.message):
                empty = True
            else:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if not empty:
            self.segments.append(self._cur_graph)
            self.segments.extend(self._seg_markers)      # bucket markers recorded inside this segment: handled after its launch
            self._seg_markers = []
        else:
            assert not self._seg_markers, "bucket markers in an empty graph segment"
            # kept alive, never replayed: destroying the only graph of a memory pool releases the pool, and the next
            # capture_begin on it trips an allocator assertion
            self._empty_graphs.append(self._cur_graph)
        self._cur_graph = None

    # ---- hipGraph capture of the whole step (static shapes): removes ~450 host launches per step ----
    def capture(self, image, gt_boxes, im_info, lr, image_offset=0, warmup=2, gt_masks=None, momentum=0.9, wd=1e-4):
        """Capture forward+backward+update into hipGraph segments (cut only at gradient all-reduces).
        The RNG step counter is read from device memory (step_dev), inputs from static buffers. momentum / wd are
        baked into the captured update kernels (lr is read from device memory: replay(lr=...)). The eager warm-up
        steps plan shapes and allocate buffers only: parameters and momentum are restored afterwards, so replay(step=0)
        is the first training step, exactly as train_step(step=0) on a fresh model would be."""
        dev = self.device
        hyper = (float(momentum), float(wd))
        self.static_in = (image.clone(), gt_boxes.clone(), im_info.clone())
        self.static_masks = gt_masks.clone() if gt_masks is not None else None
        self.step_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)   # replay(lr=...) rewrites it
        self._lr_host = lr
        snap = (self.arena.w.clone(), self.arena.m.clone()) if warmup > 0 else None
        for i in range(warmup):     # eager warm-up: plans shapes and allocates every buffer
            self.train_step(*self.static_in, step=i, image_offset=image_offset, lr=lr, gt_masks=self.static_masks,
                            momentum=hyper[0], wd=hyper[1])
        torch.cuda.synchronize()
        if snap is not None:
            self.arena.w.copy_(snap[0])
            self.arena.m.copy_(snap[1])
            self.arena.refresh_bf16()
            self.refresh_transposed()
            del snap
            torch.cuda.synchronize()
        self._pool = torch.cuda.graph_pool_handle()
        self._pool_branch = torch.cuda.graph_pool_handle()
        self._pool_opt = torch.cuda.graph_pool_handle()
        self._pool_w = torch.cuda.graph_pool_handle()
        self.segments = []
        self._empty_graphs = []
        self._buckets = []
        self._seg_markers = []
        self._deferred = []
        self._cap_opt = None
        self._final_join_opt = False
        if self.dist is not None and self._seen_buckets:
            # buckets seen in the eager warm-up: their transpose tables are built here, outside any capture
            # The per-bucket updates get no stream of their own when the RPN branch has one: HIP multiplexes streams onto
            # four hardware queues, and a fifth stream shared the main stream's queue -- every update graph (it waits for
            # its bucket's all-reduce) then blocked the data-gradient chain queued behind it (measured at world size 1:
            # 375 vs 422 img/s for the schedule without an exchange). The branch stream is idle by the time the first
            # bucket closes (the branch is joined before the RoI backward) and must wait for the updates anyway before
            # the next step's RPN branch reads the weights.
            self.opt_stream = self.branch if (self.branch is not None and os.environ.get("MXDET_TUNE_OPT_STREAM", "branch") == "branch") \
                else torch.cuda.Stream()
            self._cap_opt = (self.lr_dev,) + hyper
            for lo_hi in sorted(self._seen_buckets):
                self._transpose_table(*lo_hi)
        use_front = self.front_pipeline and self.backbone.frozen_front() > 0 and warmup > 0
        if use_front:
            from ...utils.hipgraph import GraphEvent
            # the second output buffer of the front end and every plan keyed by it (grouped launches of the consumers)
            # must exist before a capture: one more eager step on parity 1
            snap = (self.arena.w.clone(), self.arena.m.clone())
            self.backbone.eager_parity = 1
            self.train_step(*self.static_in, step=0, image_offset=image_offset, lr=lr, gt_masks=self.static_masks,
                            momentum=hyper[0], wd=hyper[1])
            self.backbone.eager_parity = 0
            torch.cuda.synchronize()
            self.arena.w.copy_(snap[0])
            self.arena.m.copy_(snap[1])
            self.arena.refresh_bf16()
            self.refresh_transposed()
            del snap
            torch.cuda.synchronize()
            self._tail_event = GraphEvent()
            fstream = self.branch if self.branch is not None else torch.cuda.Stream()
            if (self.dist is not None and getattr(self.reducer, "comm", None) is not None
                    and os.environ.get("MXDET_TUNE_FRONT_PROBE", "1") == "1"):
                fstream = self._stream_clear_of_the_exchange(fstream)
            self._front = {"graphs": [], "segments": [], "losses": [], "stream": fstream, "count": 0,
                           "ready": [torch.cuda.Event(), torch.cuda.Event()], "pool": torch.cuda.graph_pool_handle()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        for parity in ((0, 1) if use_front else (0,)):
            if use_front:
                # the front end of this parity: its own graph, captured on the front stream
                fs = self._front["stream"]
                fs.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(fs):
                    g.capture_begin(pool=self._front["pool"], capture_error_mode="thread_local")
                    c2 = self.backbone.forward_front(self.static_in[0], parity)
                    g.capture_end()
                torch.cuda.current_stream().wait_stream(fs)
                self._front["graphs"].append(g)
                self.backbone.front_override = c2
                self.segments = []
                self._buckets = []
                self._final_join_opt = False
            with torch.cuda.stream(side):
                self._cap = True
                self._seg_begin()
                self._upd, self._upd_done = (((self.lr_dev,) + hyper) if self.dist is None else None), []
                losses = self.forward_backward(*self.static_in, step=0, image_offset=image_offset, step_dev=self.step_dev,
                                               gt_masks=self.static_masks)
                self._upd = None
                self.optimizer_step(self.lr_dev, hyper[0], hyper[1])
                self._seg_end()
                if self._final_join_opt:
                    self.segments.append(("join_opt",))
                self._cap = False
                self._capture_deferred()
            if use_front:
                self._front["segments"].append(self.segments)
                self._front["losses"].append(losses)
        self.backbone.front_override = None
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_front:
            self._tail_event.record()            # the first replayed step has no predecessor to wait for
            torch.cuda.synchronize()
        self.static_losses = losses

    def _capture_deferred(self):
        """The per-bucket weight-gradient and update graphs of the exchange schedule, captured after the main capture (their
        places in the schedule are the event nodes / markers _reduce left in it)."""
        todo, self._deferred = self._deferred, []
        for d in todo:
            if d[0] == "wgrad":
                _, marker, pre, items_pre, items = d
                g = torch.cuda.CUDAGraph()
                side, self.ws.side = self.ws.side, None
                with torch.cuda.stream(side):
                    g.capture_begin(pool=self._pool_w, capture_error_mode="thread_local")
                    if pre is not None:
                        pre.side = None
                        pre.pending = items_pre
                        pre.flush()
                    self.ws.pending = items
                    self.ws.flush()
                    g.capture_end()
                self.ws.side = side
                marker[1] = g
            else:
                _, marker, lo, hi = d
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(self.opt_stream):
                    g.capture_begin(pool=self._pool_opt, capture_error_mode="thread_local")
                    self._apply_update(lo, hi, self._cap_opt, 1.0 / self.world)
                    g.capture_end()
                marker[1] = g

    def _stream_clear_of_the_exchange(self, preferred, tries=4):
        """A stream on which work is NOT held up while an all-reduce waits for its bucket's weight gradients.

        HIP multiplexes streams onto a few hardware queues (four), in order of first use, and a stream that waits for an
        event holds up every stream sharing its queue. During the weight-gradient tail of a step the communicator (its own
        stream and RCCL's internal ones) has such a wait pending most of the time; the next step's front end has to run
        beside that tail, so its stream must not share a queue with any of them (measured at world size 1: on the branch
        stream the front end started only after the step's last all-reduce, and the pipeline's gain was lost). Which queue a
        stream got cannot be asked, so it is measured: a long kernel on the weight-gradient stream, ONE all-reduce of a
        scratch tensor behind it (every rank issues the same single collective), a tiny kernel on each candidate; candidates
        whose kernel has finished while the long kernel still runs are clear. Returns `preferred` if it is clear, else the
        first clear candidate, else `preferred`."""
        import time
        comm = self.reducer.comm
        dev = self.arena.g.device
        tries = int(os.environ.get("MXDET_TUNE_FRONT_PROBE_TRIES", tries))
        while len(_PROBE_STREAMS) < tries:             # one set per process: every model's probe tries the same streams
            _PROBE_STREAMS.append(torch.cuda.Stream())
        cands = [preferred] + _PROBE_STREAMS[:tries]
        tiny = torch.zeros((64,), device=dev)
        scratch = torch.zeros((64,), device=dev)
        big = torch.empty((1 << 27,), device=dev)                    # 512 MiB: one pass ~0.2 ms
        wstream = self.ws.side if self.ws.side is not None else torch.cuda.Stream()
        for s_ in cands + [wstream]:                                  # first use of every stream: queues are bound now
            with torch.cuda.stream(s_):
                tiny.add_(0.0)
        torch.cuda.synchronize()
        done_long = torch.cuda.Event()
        with torch.cuda.stream(wstream):
            for _ in range(60):                                       # ~10 ms of work in front of the collective
                big.add_(1.0)
            done_long.record()
            ticket = comm.allreduce(scratch)                          # waits behind the long kernels: the pending wait
        evs = []
        for c in cands:
            e = torch.cuda.Event()
            with torch.cuda.stream(c):
                tiny.add_(0.0)
                e.record()
            evs.append(e)
        clear = [False] * len(cands)
        t0 = time.perf_counter()
        while not done_long.query() and time.perf_counter() - t0 < 2.0:
            for i, e in enumerate(evs):
                clear[i] = clear[i] or e.query()
            if all(clear):
                break
        ticket.wait()
        torch.cuda.synchronize()
        del big
        self.front_stream_probe = clear
        for c, ok in zip(cands, clear):
            if ok:
                return c
        return preferred

    def _mark_tail(self):
        """Called by forward_backward where the data-gradient chain has ended and only the last bucket's weight gradients,
        fold and update remain: under capture with the front-end pipeline, an event-record node the NEXT step's front end
        waits for (it then runs beside that tail: HBM-bound frozen convolutions next to MFMA-bound weight gradients)."""
        if self._cap and self._front is not None and self._tail_event is not None:
            self._tail_event.record_node()

    def replay(self, image, gt_boxes, im_info, step, gt_masks=None, lr=None):
        """One training step from the captured graphs; lr (if given) replaces the captured learning rate from here on."""
        si = self.static_in
        if lr is not None and lr != self._lr_host:
            self.lr_dev.fill_(float(lr))
            self._lr_host = lr
        segments, losses = self.segments, self.static_losses
        if self._front is not None:
            # front end of THIS batch on the front stream, behind the previous step's tail mark (not behind its end)
            fr = self._front
            par = fr["count"] & 1
            fr["count"] += 1
            fs = fr["stream"]
            if os.environ.get("MXDET_TUNE_FRONT_PIPE", "1") != "2":      # "2": no tail gate (as early as the stream allows)
                self._tail_event.wait(fs)
            with torch.cuda.stream(fs):
                if image is not si[0]:
                    si[0].copy_(image, non_blocking=True)
                if "front" not in _ABL:
                    fr["graphs"][par].replay()
                fr["ready"][par].record()
            torch.cuda.current_stream().wait_event(fr["ready"][par])
            segments, losses = fr["segments"][par], fr["losses"][par]
        elif image is not si[0]:
            si[0].copy_(image, non_blocking=True)
        if image is not si[0]:
            si[1].copy_(gt_boxes, non_blocking=True)
            si[2].copy_(im_info, non_blocking=True)
            if gt_masks is not None:
                self.static_masks.copy_(gt_masks, non_blocking=True)
        self.step_dev.fill_(step)
        fork_ev = None
        handles = {}
        for seg in segments:
            if isinstance(seg, (tuple, list)):
                if seg[0] == "wgrad":
                    seg[2].wait(self.ws.side)                 # the event node behind the producers of the bucket's dy / x
                    with torch.cuda.stream(self.ws.side):
                        seg[1].replay()
                elif seg[0] == "reduce":
                    if len(seg) > 4 and seg[4]:               # ordered behind the side stream's weight gradients
                        with torch.cuda.stream(self.ws.side):
                            h = self.reducer.reduce(seg[1], seg[2])
                    else:
                        h = self.reducer.reduce(seg[1], seg[2])
                    if len(seg) > 3:
                        handles[seg[3]] = h
                elif seg[0] == "update":
                    with torch.cuda.stream(self.opt_stream):
                        for h in handles.get(seg[2], ()):
                            h.wait()                       # orders the optimizer stream after the bucket's sums
                        seg[1].replay()
                elif seg[0] == "join_opt":
                    torch.cuda.current_stream().wait_stream(self.opt_stream)
                    if self.ws.side is not None:
                        torch.cuda.current_stream().wait_stream(self.ws.side)
                    self.reducer.pending, self.reducer.log = [], []
                elif seg[0] == "fork":
                    fork_ev = torch.cuda.Event()
                    fork_ev.record()
                elif seg[0] == "branch":
                    self.branch.wait_event(fork_ev)
                    with torch.cuda.stream(self.branch):
                        seg[1].replay()
                elif seg[0] == "join":
                    torch.cuda.current_stream().wait_stream(self.branch)
                else:
                    self.reducer.wait()
            else:
                seg.replay()
        return losses

    def optimizer_step(self, lr, momentum=0.9, wd=1e-4):
        self.ws.flush()
        self.ws.join()
        done, self._upd_done = self._upd_done, []
        if done:
            # buckets were updated as they finished; update whatever the bucket marks did not cover
            covered = sorted(done)
            gaps, pos = [], 0
            for lo, hi in covered:
                if lo > pos:
                    gaps.append((pos, lo))
                pos = max(pos, hi)
            if pos < self.arena.size:
                gaps.append((pos, self.arena.size))
            self._upd = (lr, momentum, wd)
            for lo, hi in gaps:
                self._update_range(lo, hi)
            self._upd = None
            self.ws.join()
            return
        if self._cap and self.dist is not None and self._cap_opt is not None:
            pos = 0
            for lo, hi in sorted(self._buckets):
                assert lo <= pos, "parameter range [%d, %d) belongs to no gradient bucket" % (pos, lo)
                pos = max(pos, hi)
            assert pos >= self.arena.size, "parameter range [%d, %d) belongs to no gradient bucket" % (pos, self.arena.size)
            self._final_join_opt = True      # every bucket has its own update graph; capture() appends the join
            return
        if self._cap:
            if self.dist is not None:
                self._seg_end()
                self.segments.append(("wait",))
                self._seg_begin()
        else:
            self.reducer.wait()
        rescale = 1.0 / self.world
        self.arena.sgd_step(lr, momentum, wd, rescale)
        self.refresh_transposed()

    def train_step(self, image, gt_boxes, im_info, step=0, image_offset=0, lr=0.0025, gt_masks=None,
                   momentum=0.9, wd=1e-4):
        self._upd, self._upd_done = ((lr, momentum, wd) if self.dist is None else None), []
        from ...ops import dense
        trace = not getattr(self, "_pf_wired", False) and os.environ.get("MXDET_TUNE_PREFETCH", "1") != "0"
        if trace:
            dense.PF_TRACE = []
        try:
            losses = self.forward_backward(image, gt_boxes, im_info, step, image_offset, gt_masks=gt_masks)
        finally:
            self._upd = None
            if trace:
                self._wire_prefetch(dense.PF_TRACE)
                dense.PF_TRACE = None
        self.optimizer_step(lr, momentum, wd)
        return losses

    def _wire_prefetch(self, trace):
        """Prefetch hints from the issue order of the first eager step: every single convolution launch names the filter
        the NEXT convolution launch (single or grouped, forward or data gradient) reads, so that it is warm in the
        Infinity Cache when that launch starts (mxdet_conv_desc_t.prefetch; +2 % on the step for the backbone chain
        alone). A layer launched more than once per direction (heads shared across pyramid levels) keeps its last
        successor. The order is the host's issue order: launches of the RPN branch interleave as they were issued."""
        from ...ops import dense
        for (layer, kind, w, key), (_, _, nxt, _) in zip(trace[:-1], trace[1:]):
            if nxt is None or nxt == w:          # the same filter again (a head shared across pyramid levels)
                continue
            if kind == "f":
                layer.pf_fwd = nxt
            elif kind == "b":
                layer.pf_bwd = nxt
            elif key is not None and os.environ.get("MXDET_TUNE_PREFETCH", "1") != "2":
                dense.GROUP_HINTS[key] = nxt     # grouped launch: its table is rebuilt with the hint at the next eager call
                dense._group_plans.pop(key, None)
        self._pf_wired = True
