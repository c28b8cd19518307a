"""bench.py --gpus N without torch.distributed.run: the process becomes a launcher of N rank processes and must fail
loudly (not hang, not exit 0) when the node has fewer devices."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_reports_missing_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("node has two devices: the launcher would start a real run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 visible devices" in r.stderr and "rank(s) failed" in r.stderr
    assert r.stdout.strip() == ""          # no JSON line from a failed run
