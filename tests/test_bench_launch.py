"""`python3 bench.py --gpus N` with no launcher around it starts its own ranks (the reference's sharding is one worker
per device, every worker the whole index: src/master.erl:123,140-147).  Checked here without a GPU: the launch rehearsal
(--launch-check: every rank reports who it is and leaves before anything initialises a GPU), the propagation of a failing rank's
exit code, and that a process launched by torch.distributed.run (WORLD_SIZE set) is a rank, not a launcher."""
import json
import os
import subprocess
import sys
import time

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launch_check_starts_every_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # ONE line on stdout, as for a measurement
    lc = json.loads(lines[0])["launch_check"]
    assert lc["ok"] and sorted(x["rank"] for x in lc["ranks"]) == [0, 1]
    assert all(x["world"] == 2 and x["local_rank"] == x["rank"] and x["master"].startswith("127.0.0.1:") for x in lc["ranks"])
    assert len({x["master"] for x in lc["ranks"]}) == 1        # one rendezvous for all


def test_a_failing_rank_ends_the_run_with_its_exit_code():
    """no GPU in this container: every rank leaves with the no-GPU code (3) before any rendezvous; the launcher passes it on and
    does not hang on the others"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "no GPU visible" in r.stderr and "ending the other ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 240


def test_world_size_from_a_launcher_is_respected():
    # a rank started by torch.distributed.run: rehearsal output of that rank alone, no children
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=120,
                       env=_env(RANK="2", LOCAL_RANK="2", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())["launch_check"] == {"rank": 2, "local_rank": 2, "world": 4, "master": "127.0.0.1:29999"}
    # ... and a world that disagrees with --gpus is refused
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=120,
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 2 and "must agree" in r.stderr


def test_visible_gpus_counts_without_hip(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    # (this container has no KFD topology: unknown, and the launcher then lets the ranks find out)
    assert bench.visible_gpus() is None or bench.visible_gpus() >= 0
