"""Self-started ranks (isaacgym_amd/launch.py): the path `python bench.py --gpus N` takes when no
launcher set WORLD_SIZE.  CPU only: the children are small scripts, and bench.py itself is run
once to see that a failing rank (no GPU here) ends the whole job with a non-zero code."""
import json
import os
import subprocess
import sys
import time

import pytest

from isaacgym_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = [sys.executable, "-u", os.path.join(HERE, "launch_child.py")]


def test_rank_env_is_what_torchrun_sets():
    e = launch.rank_env(2, 4, 1234, base={})
    assert e["RANK"] == "2" and e["LOCAL_RANK"] == "2" and e["WORLD_SIZE"] == "4"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "1234"
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_relays_rank0_only():
    rc, out = launch.spawn_ranks(CHILD + ["ok"], 3, timeout=60)
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["local_rank"] == 0 and d["addr"] == "127.0.0.1"


def test_failed_rank_fails_the_job_and_stops_the_others():
    t0 = time.monotonic()
    rc, _ = launch.spawn_ranks(CHILD + ["fail"], 3, timeout=50)
    assert rc == 7
    assert time.monotonic() - t0 < 30  # the sleeping ranks were stopped, not waited for


def test_timeout_stops_everything():
    t0 = time.monotonic()
    rc, _ = launch.spawn_ranks(CHILD + ["hang"], 2, timeout=1.0)
    assert rc == 124
    assert time.monotonic() - t0 < 30


def test_ranks_rendezvous_over_gloo():
    rc, out = launch.spawn_ranks(CHILD + ["gloo"], 2, timeout=120)
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1])["sum"] == 3.0


def test_bench_without_launcher_fails_cleanly_without_gpus():
    """No GPU in this container: every rank dies at set_device, and the parent must report that."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--spawn-timeout", "120"], env=env, capture_output=True, text=True, timeout=200)
    assert p.returncode != 0
    assert "rank" in p.stderr and "exited with code" in p.stderr
    assert not p.stdout.strip()


def test_bench_group_rccl_failure_fails_every_rank():
    """RCCL cannot come up here (no GPU): without --allow-gloo-fallback every rank leaves with code 3."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the GPU suite triggers the refusal with two ranks on one device")
    rc, out = launch.spawn_ranks(CHILD + ["bench_group"], 2, timeout=150)
    assert rc == 3
    assert "{" not in out  # no result line


def test_bench_group_fallback_names_gloo():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out = launch.spawn_ranks(CHILD + ["bench_group_fallback"], 2, timeout=150)
    assert rc == 0
    d = json.loads(out.strip().splitlines()[-1])
    assert d["backend"] == "gloo" and d["rccl_ranks"] == 0 and "RCCL bring-up failed" in d["note"] and d["sum"] == 3.0
