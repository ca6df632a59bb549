"""The learner-facing wrappers (RLGPUEnv, get_rlgames_env_creator, MultiObserver) on host-only stand-ins."""
import pytest

from isaacgym_amd import tasks
from isaacgym_amd.utils import rlgames_utils as R


class FakeTask:
    num_states = 0
    num_envs = 128

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture, force_render,
                 **kw):
        self.cfg, self.rl_device, self.sim_device, self.kw = cfg, rl_device, sim_device, kw
        self.num_states = cfg["env"].get("numStates", 0)
        self.calls = []

    action_space = dict(shape=(8,))
    observation_space = dict(shape=(60,))
    state_space = dict(shape=(5,))

    def step(self, actions):
        self.calls.append(("step", actions))
        return "obs", "rew", "done", "info"

    def reset(self):
        return "first obs"

    def reset_done(self):
        return "obs after resets"

    def get_number_of_agents(self):
        return 1

    def set_train_info(self, env_frames, *a, **k):
        self.calls.append(("frames", env_frames))


@pytest.fixture
def fake_task(monkeypatch):
    monkeypatch.setitem(tasks.isaacgym_task_map, "Fake", FakeTask)


def test_env_creator_maps_rank_to_device_and_offsets_the_draws(fake_task, monkeypatch):
    cfg = {"env": {"numEnvs": 128}}
    hooked = []
    make = R.get_rlgames_env_creator(7, cfg, "Fake", "cuda:0", "cuda:0", multi_gpu=True, post_create_hook=lambda: hooked.append(1),
                                     sim="engine")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("RANK", "11")
    monkeypatch.setenv("WORLD_SIZE", "16")
    env = make()
    assert (env.rl_device, env.sim_device) == ("cuda:3", "cuda:3") and cfg["rank"] == 3 and cfg["rl_device"] == "cuda:3"
    assert env.env_offset == 11 * 128 and env.kw == {"sim": "engine", "seed": 7} and hooked == [1]
    assert R.multi_gpu_get_rank(True) == 3 and R.multi_gpu_get_rank(False) == 0
    single = R.get_rlgames_env_creator(7, {"env": {}}, "Fake", "cuda:1", "cpu")()
    assert (single.rl_device, single.sim_device, single.env_offset) == ("cpu", "cuda:1", 0)


def test_rlgpu_env_passes_through(fake_task):
    R.env_configurations.register("rlgpu-test", {"vecenv_type": "RLGPU",
                                                 "env_creator": lambda **kw: FakeTask({"env": kw}, "cpu", "cpu", 0, True, False, False)})
    env = R.RLGPUEnv("rlgpu-test", 1, numStates=5)
    assert env.step("a") == ("obs", "rew", "done", "info") and env.reset() == "first obs"
    assert env.reset_done() == "obs after resets" and env.get_number_of_agents() == 1
    assert env.get_env_info() == {"action_space": dict(shape=(8,)), "observation_space": dict(shape=(60,)),
                                  "state_space": dict(shape=(5,))}
    assert "state_space" not in R.RLGPUEnv("rlgpu-test", 1).get_env_info()
    env.set_train_info(1000)
    assert env.env.calls[-1] == ("frames", 1000) and env.get_env_state() is None
    env.set_env_state({"x": 1})  # the task has no such hook: ignored, as in the reference


def test_multi_observer_fans_out():
    class Obs:
        def __init__(self):
            self.seen = []

        def __getattr__(self, name):
            return lambda *a, **k: self.seen.append((name, a))
    a, b = Obs(), Obs()
    m = R.MultiObserver([a, b])
    m.before_init("base", {}, "exp")
    m.after_init("algo")
    m.process_infos({"k": 1}, [0])
    m.after_steps()
    m.after_clear_stats()
    m.after_print_stats(10, 2, 1.5)
    assert a.seen == b.seen and [n for n, _ in a.seen] == ["before_init", "after_init", "process_infos", "after_steps",
                                                            "after_clear_stats", "after_print_stats"]
    assert a.seen[-1][1] == (10, 2, 1.5)


def test_set_seed_follows_the_reference_rule():
    import random

    import numpy as np
    import torch

    from isaacgym_amd.utils import utils as U
    assert U.resolve_seed(-1, True, 3) == 45 and U.resolve_seed(7, False, 2) == 9 and 0 <= U.resolve_seed(-1, False) < 10000
    try:
        assert U.set_seed(5, rank=1) == 6
        a = (random.random(), float(np.random.rand()), float(torch.rand(())))
        assert U.set_seed(6) == 6
        assert (random.random(), float(np.random.rand()), float(torch.rand(()))) == a
        assert U.flatten_dict({"a": {"b": 1}, "c": 2}) == {"a.b": 1, "c": 2}
    finally:
        torch.use_deterministic_algorithms(False)
