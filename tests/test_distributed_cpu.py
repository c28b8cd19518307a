"""N > 1 path on CPU: two gloo ranks exchange gradient buckets exactly as the GPU ranks do over RCCL."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mxdetection_amd.models.utils.dp import BucketReducer
    n = 1_000_003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1) * 1e-3
    red = BucketReducer(g, dist, max_bucket_elems=200_000)
    marks = [0, 123_456, 123_456, 700_001, n]        # includes an empty bucket
    for lo, hi in zip(marks[:-1], marks[1:]):
        red.reduce(lo, hi)
    log = red.wait()
    want = torch.arange(n, dtype=torch.float32) * 1e-3 * sum(r + 1 for r in range(world))
    ok = torch.allclose(g, want, rtol=1e-6)
    covered = sorted(log)
    contiguous = covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))
    small = all(e - s <= 200_000 for s, e in covered)
    # weights start identical on every rank: broadcast from rank 0
    w = torch.full((10,), float(rank))
    dist.broadcast(w, 0)
    torch.save({"ok": bool(ok), "contiguous": contiguous, "small": small, "w": w, "g0": g[:4].clone()},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    for r in res:
        assert r["ok"] and r["contiguous"] and r["small"]
        assert torch.all(r["w"] == 0)
    assert torch.equal(res[0]["g0"], res[1]["g0"])      # every rank ends with the same sums


def test_ranks_get_disjoint_images_and_rng_streams():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    # per-rank image offsets never overlap, so Philox sampling keys (seed, step, image) are rank-unique
    offs = [set(range(r * bench.BATCH_PER_GPU, (r + 1) * bench.BATCH_PER_GPU)) for r in range(8)]
    assert all(offs[i].isdisjoint(offs[j]) for i in range(8) for j in range(i + 1, 8))
    a = bench.synth_batch(0, 0, "cpu")
    b = bench.synth_batch(1, 0, "cpu")
    assert a[0].shape == (bench.BATCH_PER_GPU, 3, 800, 1344) and not torch.equal(a[0], b[0])
    assert torch.all(a[0][..., 1333:] == 0)
    assert a[1].shape == (bench.BATCH_PER_GPU, 100, 5)
    valid = a[1][..., 4] >= 0
    assert 4 <= int(valid[0].sum()) <= 16
    bx = a[1][valid]
    assert torch.all(bx[:, 2] <= 1332.0 + 1e-3) and torch.all(bx[:, 3] <= 799.0 + 1e-3) and torch.all(bx[:, :2] >= 0)
