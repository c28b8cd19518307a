"""datasets/loader (/root/reference/README.md:22): per-rank batch loader in front of the training step.

MXNet-lineage role: `AnchorLoader` -- shuffle with aspect-ratio grouping, cut the epoch into batches, split each batch
over the contexts, run the cv2 transforms on the host and stack zero-padded NCHW float tensors. Redesign for one
process per GPU:
  * the epoch order is a pure function of (seed, epoch) -- every rank computes the same order and takes its own slice
    of each global batch, so there is no sampler state to exchange (images are the shard unit);
  * the host only READS frames (worker threads) into a pinned staging slot; one H2D copy per batch moves the raw
    8-bit frames (about 1/7 of the bytes of the fp32 tensor the lineage uploads), and flip / resize / normalise / pad
    happen in one kernel on a copy stream (process_data.BatchPreprocessor), double-buffered against the training step;
  * ground-truth boxes are transformed on the host (a few floats) and padded to [N, g_max, 5] with -1, the layout the
    target-assignment kernels take; instance masks are rasterised from polygons on the device at network resolution.
"""
import queue
import threading

import numpy as np

from ..process_data import transform as T


def epoch_order(roidb, global_batch, epoch, seed=0, shuffle=True, aspect_grouping=True):
    """Image indices of one epoch, a multiple of global_batch long (wrapped), identical on every rank.

    With aspect_grouping, landscape and portrait images never share a batch (less padding): both groups are permuted,
    cut into batches, and the batches are permuted (lineage `AnchorLoader.reset`)."""
    n = len(roidb)
    if n == 0:
        return np.zeros((0,), np.int64)
    rng = np.random.default_rng([seed, epoch])
    if not shuffle:
        order = np.arange(n)
    elif aspect_grouping:
        horz = np.array([r["width"] >= r["height"] for r in roidb])
        groups = []
        for sel in (np.where(horz)[0], np.where(~horz)[0]):
            sel = rng.permutation(sel)
            if sel.size % global_batch:                      # fill the group's last batch from its own start, cyclically
                sel = np.resize(sel, sel.size + global_batch - sel.size % global_batch)   # (the group may be smaller than the pad)
            groups.append(sel.reshape(-1, global_batch))
        rows = np.concatenate(groups, 0)
        order = rows[rng.permutation(rows.shape[0])].reshape(-1)
    else:
        order = rng.permutation(n)
    if order.size % global_batch:
        order = np.resize(order, order.size + global_batch - order.size % global_batch)   # cyclic: n may be < the pad
    return order.astype(np.int64)


class HostBatch:
    """What the host side produces for one per-rank batch (numpy only; no device calls)."""
    __slots__ = ("frames", "flips", "shapes", "scales", "resized", "pad", "gt", "im_info", "poly", "indices")


class DetectionLoader:
    def __init__(self, roidb, batch_per_gpu=2, device="cuda", rank=0, world=1, reader=None, preprocessor=None,
                 g_max=100, with_masks=False, shuffle=True, aspect_grouping=True, seed=0, num_workers=4, depth=2):
        from .synthetic import synthetic_reader
        self.roidb, self.b, self.device = roidb, batch_per_gpu, device
        self.rank, self.world = rank, world
        self.reader = reader or synthetic_reader
        self.pre = preprocessor or T.BatchPreprocessor(pad_to="orient")
        self.g_max, self.with_masks = g_max, with_masks
        self.shuffle, self.aspect_grouping, self.seed = shuffle, aspect_grouping, seed
        self.num_workers, self.depth = max(1, num_workers), max(2, depth)
        self.epoch = 0
        self._slots = None

    # ---- host side (testable without a GPU) ---------------------------------------------------------------------
    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        gb = self.b * self.world
        return len(epoch_order(self.roidb, gb, self.epoch, self.seed, self.shuffle, self.aspect_grouping)) // gb

    def rank_batches(self):
        """[num_batches, batch_per_gpu] roidb indices of THIS rank for the current epoch."""
        gb = self.b * self.world
        order = epoch_order(self.roidb, gb, self.epoch, self.seed, self.shuffle, self.aspect_grouping)
        return order.reshape(-1, gb)[:, self.rank * self.b:(self.rank + 1) * self.b]

    def assemble(self, indices):
        """Read the frames and build the ground truth of one batch."""
        hb = HostBatch()
        entries = [self.roidb[int(i)] for i in indices]
        hb.indices = [int(i) for i in indices]
        hb.frames = [np.ascontiguousarray(self.reader(e), dtype=np.uint8) for e in entries]
        hb.flips = [bool(e.get("flipped", False)) for e in entries]
        hb.shapes = [(f.shape[0], f.shape[1]) for f in hb.frames]
        hb.scales, hb.resized, hb.pad = self.pre.plan(hb.shapes)
        N = len(entries)
        hb.gt = -np.ones((N, self.g_max, 5), np.float32)
        hb.im_info = np.zeros((N, 3), np.float32)
        polys = []
        for n, e in enumerate(entries):
            G = min(int(e["boxes"].shape[0]), self.g_max)
            w = hb.shapes[n][1]
            bx = T.transform_boxes(e["boxes"][:G], hb.scales[n], hb.flips[n], w)
            # keep boxes inside the resized frame (scale rounding can push x2 one pixel out)
            if G:
                bx[:, 0::2] = np.clip(bx[:, 0::2], 0, hb.resized[n][1] - 1)
                bx[:, 1::2] = np.clip(bx[:, 1::2], 0, hb.resized[n][0] - 1)
            hb.gt[n, :G, :4] = bx
            hb.gt[n, :G, 4] = e["gt_classes"][:G]
            hb.im_info[n] = (hb.resized[n][0], hb.resized[n][1], hb.scales[n])
            if self.with_masks:
                polys.append([T.transform_polygons(p, hb.scales[n], hb.flips[n], w) for p in e.get("polygons", [])[:G]])
        hb.poly = T.pack_polygons(polys, N, self.g_max) if self.with_masks else None
        return hb

    # ---- device side ---------------------------------------------------------------------------------------------
    def _make_slot(self):
        import torch
        s = {}
        s["pinned"] = torch.empty((8 << 20,), dtype=torch.uint8, pin_memory=True)
        s["raw"] = torch.empty((8 << 20,), dtype=torch.uint8, device=self.device)
        s["gt_pin"] = torch.empty((self.b, self.g_max, 5), dtype=torch.float32, pin_memory=True)
        s["gt"] = torch.empty((self.b, self.g_max, 5), dtype=torch.float32, device=self.device)
        s["info_pin"] = torch.empty((self.b, 3), dtype=torch.float32, pin_memory=True)
        s["info"] = torch.empty((self.b, 3), dtype=torch.float32, device=self.device)
        s["img"] = None
        s["masks"] = None
        s["ready"] = torch.cuda.Event()
        s["free"] = torch.cuda.Event()
        s["free"].record()
        s["ready"].record()
        return s

    def _upload(self, hb, s):
        """Enqueue H2D + preprocess of one host batch into device slot s on the copy stream."""
        import torch
        total = sum(f.size for f in hb.frames)
        if s["pinned"].numel() < total:
            s["pinned"] = torch.empty((total * 5 // 4,), dtype=torch.uint8, pin_memory=True)
            s["raw"] = torch.empty((total * 5 // 4,), dtype=torch.uint8, device=self.device)
        s["ready"].synchronize()      # the slot's previous H2D has left the pinned buffer (long done in steady state)
        stage = s["pinned"].numpy()
        offs, o = [], 0
        for f in hb.frames:
            stage[o:o + f.size] = f.reshape(-1)
            offs.append(o)
            o += f.size
        s["gt_pin"].numpy()[...] = hb.gt
        s["info_pin"].numpy()[...] = hb.im_info
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(s["free"])            # the step that last read this slot has been enqueued
            s["raw"][:total].copy_(s["pinned"][:total], non_blocking=True)
            s["gt"].copy_(s["gt_pin"], non_blocking=True)
            s["info"].copy_(s["info_pin"], non_blocking=True)
            frames = [s["raw"][offs[n]:offs[n] + f.size].view(f.shape[0], f.shape[1], 3) for n, f in enumerate(hb.frames)]
            hp, wp = hb.pad
            if s["img"] is None or tuple(s["img"].shape) != (len(frames), 3, hp, wp):
                s["img"] = torch.empty((len(frames), 3, hp, wp), dtype=torch.bfloat16, device=self.device)
            self.pre(frames, hb.flips, out=s["img"])
            if self.with_masks:
                v, ps, inf = [torch.from_numpy(a).to(self.device, non_blocking=False) for a in hb.poly]
                if s["masks"] is None or tuple(s["masks"].shape) != (len(frames), self.g_max, hp, wp):
                    s["masks"] = torch.empty((len(frames), self.g_max, hp, wp), dtype=torch.uint8, device=self.device)
                if v.numel() == 0:
                    v = torch.zeros((1, 2), dtype=torch.float32, device=self.device)
                T.polygon_masks(v, ps, inf, len(frames), self.g_max, hp, wp, out=s["masks"])
            s["ready"].record(self._copy_stream)

    def __iter__(self):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        if self._slots is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._slots = [self._make_slot() for _ in range(self.depth)]
        batches = self.rank_batches()
        host_q = queue.Queue(maxsize=self.depth + 1)
        stop = threading.Event()

        def produce():
            # a failing reader must not leave the consumer waiting on the queue: the exception travels through it
            try:
                with ThreadPoolExecutor(self.num_workers) as pool:
                    pending = []
                    it = iter(batches)
                    for _ in range(self.num_workers):
                        nxt = next(it, None)
                        if nxt is None:
                            break
                        pending.append(pool.submit(self.assemble, nxt))
                    while pending and not stop.is_set():
                        hb = pending.pop(0).result()
                        nxt = next(it, None)
                        if nxt is not None:
                            pending.append(pool.submit(self.assemble, nxt))
                        while not stop.is_set():
                            try:
                                host_q.put(hb, timeout=0.1)
                                break
                            except queue.Full:
                                pass
                    for f in pending:
                        f.cancel()
                item = None
            except BaseException as ex:  # noqa: BLE001
                item = ex
            while not stop.is_set():
                try:
                    host_q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    pass

        def take():
            item = host_q.get()
            if isinstance(item, BaseException):
                raise RuntimeError("the loader's reader thread failed") from item
            return item

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            k, inflight = 0, []
            hb = take()
            while True:
                # slots outside `inflight` are free: keep every one of them uploading ahead of the batch handed out
                while hb is not None and len(inflight) < self.depth:
                    s = self._slots[k % self.depth]
                    self._upload(hb, s)
                    inflight.append(s)
                    k += 1
                    hb = take()
                if not inflight:
                    break
                s = inflight.pop(0)
                cur = torch.cuda.current_stream()
                cur.wait_event(s["ready"])
                out = {"image": s["img"], "gt_boxes": s["gt"], "im_info": s["info"]}
                if self.with_masks:
                    out["gt_masks"] = s["masks"]
                yield out
                s["free"].record(cur)        # everything the consumer enqueued on this slot precedes the next upload
        finally:
            stop.set()
            while th.is_alive():
                try:
                    host_q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
