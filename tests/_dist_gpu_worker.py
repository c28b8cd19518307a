"""Worker of tests/test_gpu_dist2.py: one data-parallel rank (gloo over CUDA tensors, both ranks on the one GPU of
the test box). Not a test module itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    from test_gpu_model import _inputs
    from mxdetection_amd.models import FasterRCNN
    backend = os.environ.get("MXDET_TEST_BACKEND", "gloo")
    if backend == "nccl":                      # RCCL wants one device per rank
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    N, H, W = 2, 256, 320
    m = FasterRCNN("cuda", seed=7, pre_nms_top_n=1000, post_nms_top_n=1000)
    m.enable_wgrad_stream()
    m.enable_branch_stream()
    m.enable_grouped_wgrad()
    m.enable_data_parallel(world)
    if backend == "nccl":          # two devices: the exchange must go through the library's own collective (csrc/comm.hip)
        assert m.comm is not None, "RCCL process group without the C-ABI communicator"
    dist.broadcast(m.arena.w, 0)
    m.arena.refresh_bf16()
    m.refresh_transposed()
    batch = _inputs(N, H, W, seed=10 + rank)
    m.train_step(*batch, step=0, image_offset=rank * N, lr=0.001)     # training step 0, eager (bucketed all-reduce)
    m.capture(*batch, lr=0.001, image_offset=rank * N, warmup=1)      # the capture's own warm-up leaves the weights alone
    losses = torch.cat(m.replay(*batch, 1)).clone()                   # step 1 from the captured graphs
    torch.cuda.synchronize()
    w = m.arena.w.detach().cpu().numpy()
    np.savez(out, sample=w[::97].copy(), checksum=np.array([np.float64(w.astype(np.float64).sum())]),
             losses=losses.cpu().numpy(), nbuckets=np.array([len(m._seen_buckets)]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
