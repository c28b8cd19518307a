"""ops: batched bitmask NMS (MXNet role: contrib.box_nms / cpu_nms / gpu_nms)."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def nms_batched(boxes, counts, thresh, max_keep=None, invalid=None):
    """boxes [B,n,4] f32 sorted by descending score, counts [B] i32. Returns (keep_idx [B,n] i32, num_keep [B])."""
    lib = _lib.load()
    B, n = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    keep = torch.full((B, max(n, 1)), -1, dtype=torch.int32, device=dev)
    num = torch.zeros((B,), dtype=torch.int32, device=dev)
    ws_bytes = lib.mxdet_nms_batched_workspace_bytes(B, n)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=dev)
    check(lib.mxdet_nms_batched(ptr(boxes), ptr(counts), ptr(invalid), B, n, thresh,
                                n if max_keep is None else max_keep, ptr(keep), ptr(num), ptr(ws), ws_bytes,
                                stream_ptr()), "nms_batched")
    return keep, num
