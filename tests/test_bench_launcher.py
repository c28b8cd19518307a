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


def _sleeper(tmp_path):
    f = tmp_path / "sleeper.py"
    f.write_text("import os, sys, time\nopen(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    return str(f)


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie still answers kill(0): look at its state
    try:
        with open("/proc/%d/stat" % pid) as fh:
            return fh.read().split(") ")[1][0] != "Z"
    except OSError:
        return False


def _wait_pids(stem, n, timeout=20):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(os.path.exists("%s.%d" % (stem, r)) and open("%s.%d" % (stem, r)).read() for r in range(n)):
            return [int(open("%s.%d" % (stem, r)).read()) for r in range(n)]
        time.sleep(0.1)
    raise AssertionError("ranks did not start")


def test_launcher_deadline_stops_its_ranks(tmp_path):
    """No GPU involved: the ranks are sleepers. The deadline passes -> exit code 124, no rank left behind."""
    sys.path.insert(0, ROOT)
    import bench
    stem = str(tmp_path / "pid")
    rc = bench.spawn_ranks(2, deadline_s=1.5, child_cmd=[sys.executable, _sleeper(tmp_path), stem])
    assert rc == 124
    import time
    time.sleep(0.3)
    assert not any(_alive(p) for p in _wait_pids(stem, 2))


def test_killed_launcher_takes_its_ranks_along(tmp_path):
    """A driver timeout kills only the launcher (SIGTERM, then SIGKILL): the ranks must not stay behind holding the GPUs."""
    import signal
    import time
    stem = str(tmp_path / "pid")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, child_cmd=[sys.executable, %r, %r]))" % (ROOT, _sleeper(tmp_path), stem))
    for sig in (signal.SIGTERM, signal.SIGKILL):
        for r in range(2):
            if os.path.exists("%s.%d" % (stem, r)):
                os.remove("%s.%d" % (stem, r))
        launcher = subprocess.Popen([sys.executable, "-c", code])
        pids = _wait_pids(stem, 2)
        launcher.send_signal(sig)
        launcher.wait(timeout=40)
        t0 = time.time()
        while time.time() - t0 < 10 and any(_alive(p) for p in pids):
            time.sleep(0.1)
        assert not any(_alive(p) for p in pids), "ranks survived the launcher (signal %d)" % sig
