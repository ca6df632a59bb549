"""Two ranks on one GPU over gloo, started by isaacgym_amd.launch: the product's shard entry points
(AllegroHand / ShadowHand shard kernels + queued records, the sharded quantile) against the unsharded calls."""
import json
import os
import sys

import pytest

from isaacgym_amd import launch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_on_one_gpu_equal_the_unsharded_calls():
    rc, out = launch.spawn_ranks([sys.executable, "-u", os.path.join(HERE, "shard_child.py")], 2, timeout=110)
    assert rc == 0, out
    d = json.loads(out.strip().splitlines()[-1])
    assert set(d) == {"allegro", "shadow", "quantile", "window_quantile", "ant_imitate"}


def test_rccl_paths_on_one_rank():
    """The collectives of the package through the nccl (= RCCL) backend on device tensors - one rank is all a one-GPU
    box offers, so the world-of-one shortcuts are switched off: histogram all-reduce and slot all-gather of the
    two-collective quantile, the queued shard records on their side stream."""
    tool = os.path.join(os.path.dirname(HERE), "tools", "nccl_one_rank_check.py")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    rc, out = launch.spawn_ranks([sys.executable, "-u", tool], 1, timeout=110, env=env)
    assert rc == 0, out
    assert "window quantile over nccl ok, collectives per call: 2" in out and "queued shard records over nccl ok" in out


# ---- bench.py's N > 1 path on a one-GPU box ------------------------------------------------------------
BENCH = os.path.join(os.path.dirname(HERE), "bench.py")
SMALL = ["--steps", "3", "--warmup", "1", "--envs-per-gpu", "65536", "--no-cpu-baseline", "--spawn-timeout", "100"]


def _bench(*extra):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH] + SMALL + list(extra), env=env, capture_output=True, text=True, timeout=200)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p.returncode, lines, p.stderr


def test_bench_rccl_refusal_fails_the_run():
    """Two ranks on ONE device: RCCL refuses the communicator ("Duplicate GPU detected").  A result line that was
    quietly measured over another backend would be worthless as a scaling record, so the run must fail."""
    rc, lines, err = _bench("--gpus", "2", "--all-ranks-on-device", "0", "--no-configs")
    assert rc != 0, (lines, err[-2000:])
    assert not lines, lines
    assert "RCCL" in err


def test_bench_fallback_flag_names_gloo():
    rc, lines, err = _bench("--gpus", "2", "--all-ranks-on-device", "0", "--no-configs", "--allow-gloo-fallback")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["collectives"]["barrier"] == "gloo" and d["collectives"]["max_of_elapsed"] == "gloo"
    assert "RCCL bring-up failed" in d["collectives"]["note"]


def test_bench_world_of_one_through_rccl_with_configs():
    """--force-group: the barrier and the MAX of the N > 1 path through RCCL as a world of one; the other
    configurations of the path ride along in `configs`."""
    rc, lines, err = _bench("--force-group", "--config-steps", "3")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["rccl_ranks"] == 1 and d["config"]["collective_backend"] == "nccl" and d["collectives"]["barrier"] == "nccl"
    assert d["roofline"]["traffic"] is None  # the committed PMC figure belongs to the 2^20-env run only
    for name in ("humanoid_step", "allegro_post", "allegro_pre", "humanoid_amp_step", "motion_amp_obs_1024",
                 "motion_amp_obs_1048576", "amp_demo_fetch_device_sampling", "amp_demo_fetch_streamed_to_host",
                 "torch_jit_utils.quat_mul", "torch_jit_utils.tensor_clamp", "cartpole_step", "shadow_post",
                 "humanoid_imitate_step", "ant_imitate_step", "humanoid_amp_reset_masked"):
        assert d["configs"][name]["us"] > 0, name
    assert d["configs"]["humanoid_step"]["bytes_per_unit"] == 940 and "frac" in d["configs"]["humanoid_amp_step"]


def test_bench_two_ranks_with_configs_run_the_sharded_humanoid_configuration():
    """Two ranks (on one device, so over the gloo fallback): the configurations ride along on every rank, the
    slowest rank's time is reported, and the Humanoid shard runs the episode-statistics kernel per step with the
    totals all-gathered over the ranks (the block that only a world larger than one executes)."""
    rc, lines, err = _bench("--gpus", "2", "--all-ranks-on-device", "0", "--allow-gloo-fallback", "--config-steps", "3")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * 65536
    c = d["configs"]["humanoid_step+episode_stats"]
    assert c["us"] > 0 and c["bytes_per_unit"] == 940 + 28 and "2 ranks" in c["what"] and "over gloo" in c["what"]
    assert d["collectives"]["episode_stats_all_gather"] == "gloo"
    g = d["configs"]["amp_demo_fetch+all_gather"]
    assert g["us"] > 0 and g["units"] == 2 * 1024 and "1024 rows per gathered demo buffer" in g["what"], g
    assert d["collectives"]["imitation_batch_all_gather"] == "gloo"
    for name in ("humanoid_step", "allegro_post", "humanoid_amp_step", "torch_jit_utils.slerp", "cartpole_step",
                 "shadow_post", "humanoid_imitate_step", "ant_imitate_step", "motion_amp_obs_1024"):
        assert d["configs"][name]["us"] > 0, name
