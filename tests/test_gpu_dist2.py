"""Two data-parallel ranks end to end on the GPU box: the N > 1 schedule (eager warm-up step with bucketed
all-reduce, then hipGraph segments cut at every bucket + per-bucket update graphs) against a single-process
restatement that averages the two ranks' gradients by hand. The box has one GPU, so both ranks share it and the
exchange goes over gloo (CUDA tensors); RCCL itself is exercised at world size 1 in test_gpu_model.py and by the
driver's multi-GPU run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_match_hand_averaged_reference(hip, tmp_path):
    import torch
    from test_gpu_model import _inputs
    from mxdetection_amd.models import FasterRCNN
    from mxdetection_amd.ops import dense
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    # two visible devices: the exchange goes over RCCL, one rank per device; one device: gloo on CUDA tensors
    # (RCCL refuses two ranks on one device), i.e. "RCCL N>1 unverified" on such a box
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    print("exchange backend:", backend)
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MXDET_TEST_BACKEND=backend)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(r), "2", str(port), outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=500)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    assert np.array_equal(r0["sample"], r1["sample"]) and r0["checksum"][0] == r1["checksum"][0]   # replicas stay identical
    assert int(r0["nbuckets"][0]) >= 4                                       # the exchange really was bucketed
    assert not np.array_equal(r0["losses"], r1["losses"])                    # different images per rank

    # single-process restatement: both ranks' batches per step, gradients averaged by hand, one SGD-momentum update
    N, H, W = 2, 256, 320
    m = FasterRCNN("cuda", seed=7, pre_nms_top_n=1000, post_nms_top_n=1000)
    w0 = m.arena.w.clone()
    batches = [_inputs(N, H, W, seed=10 + r) for r in range(2)]
    for step in (0, 1):
        gsum = torch.zeros_like(m.arena.g)
        for r in range(2):
            m.forward_backward(*batches[r], step=step, image_offset=r * N)
            torch.cuda.synchronize()
            gsum += m.arena.g
        dense.sgd_momentum_update(m.arena.w, gsum, m.arena.m, m.arena.wb, 0.001, 0.9, 1e-4, 0.5)
        m.refresh_transposed()
    torch.cuda.synchronize()
    ref = m.arena.w.cpu().numpy()[::97]
    moved = float(np.abs(ref - w0.cpu().numpy()[::97]).max())
    assert moved > 0
    # proposals rank tied bf16 logits: the last bit of a weight can reorder them between the two schedules, so the
    # second step is compared with the tolerance the world-1 test uses
    assert float(np.abs(ref - r0["sample"]).max()) <= 1e-2 * moved
