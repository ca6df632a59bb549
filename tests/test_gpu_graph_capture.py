"""The step entry points enqueue kernels on the caller's stream and nothing else: no
allocation, no copy through the host, no synchronisation.  The proof is that they can be
recorded into a HIP graph on a capturing stream and that replays of the graph on new inputs
give the bits of an eager call on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from isaacgym_amd import _abi, ops, synthetic  # noqa: E402

DEV = "cuda:0"


def _clone(t):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}


def _same_bits(a, b, keys):
    for k in keys:
        if torch.is_tensor(a.get(k)):
            x, y = a[k], b[k]
            assert torch.equal(x.view(torch.uint8) if x.dtype != torch.bool else x,
                               y.view(torch.uint8) if y.dtype != torch.bool else y), k


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # the first call loads the code object: keep that out of the capture
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


@pytest.mark.parametrize("spec,step", [(synthetic.ANT, ops.ant_step), (synthetic.HUMANOID, ops.humanoid_step)])
def test_loco_step_replays_from_a_graph(spec, step):
    n = 20011
    p = synthetic.loco_params(spec, seed=8, rng_offset=5)
    static = synthetic.loco_state(spec, n, 1, DEV, with_rand=True)  # caller-supplied uniforms: nothing baked into the launch
    static["obs_clamped"] = torch.zeros(n, spec["num_obs"], device=DEV)
    static["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    p.clip_obs = 1.5
    scratch = _clone(static)
    g = _capture(lambda: step(p, static, n))
    for s in range(3):
        fresh = synthetic.loco_state(spec, n, 50 + s, DEV, with_rand=True)
        fresh["reset"][::7] = 1
        for k, v in fresh.items():
            static[k].copy_(v)
            scratch[k].copy_(v)
        g.replay()
        step(p, scratch, n)
        torch.cuda.synchronize()
        _same_bits(static, scratch, static.keys())


def test_hand_post_physics_with_its_reduction_replays_from_a_graph():
    n = 9001
    p = synthetic.allegro_params("full_state", seed=2, rng_offset=1)
    static = synthetic.allegro_state(n, 5, DEV, "full_state", True)
    scratch = _clone(static)
    g = _capture(lambda: ops.allegro_post_physics(p, static, n))
    for s in range(2):
        fresh = synthetic.allegro_state(n, 60 + s, DEV, "full_state", True)
        for k, v in fresh.items():
            if torch.is_tensor(v):
                static[k].copy_(v)
                scratch[k].copy_(v)
        g.replay()
        ops.allegro_post_physics(p, scratch, n)
        torch.cuda.synchronize()
        _same_bits(static, scratch, [k for k in static if k != "workspace"])


def test_quantile_select_replays_from_a_graph():
    n = 70001
    x = torch.zeros(n, dtype=torch.float64, device=DEV)
    ws = torch.empty(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    out = {}

    def run():
        out["q"] = ops.quantile_f64(x, 0.9, ws)

    g = _capture(run)
    gen = torch.Generator().manual_seed(3)
    for s in range(3):
        v = torch.rand(n, dtype=torch.float64, generator=gen) * (s + 1)
        x.copy_(v)
        g.replay()
        torch.cuda.synchronize()
        assert float(out["q"]) == float(torch.quantile(v, 0.9))


@pytest.mark.parametrize("name,draws_per_step", [("Cartpole", 1), ("Ant", 1), ("Humanoid", 1), ("AntImitate", 1),
                                                  ("HumanoidImitate", 1), ("AllegroHand", 2), ("ShadowHand", 2)])
def test_whole_task_step_replays_from_a_graph(name, draws_per_step):
    """task.step recorded once (isaacgym_amd.graph.GraphedStep) against an eager twin: same
    observations, rewards, flags and reset draws, bit for bit, over steps that reset environments."""
    from isaacgym_amd import tasks
    from isaacgym_amd.graph import GraphedStep
    n, warm = 3000, 2
    eager, graphed = tasks.make(name, n, DEV, seed=3), tasks.make(name, n, DEV, seed=3)
    for t in (eager, graphed):
        t.sim._sets = t.sim._sets[:1]  # one set of simulator tensors, rewritten in place below
    gs = GraphedStep(graphed, warmup=warm)
    zero = torch.zeros(n, eager.num_actions, device=DEV)
    for _ in range(warm):
        eager.step(zero)
    gen = torch.Generator(device=DEV).manual_seed(5)
    fresh = tasks.make(name, n, DEV, seed=77).sim._sets  # other plausible simulator states to rotate in
    for i in range(1, 5):
        for k, v in fresh[i % len(fresh)].items():
            eager.sim._sets[0][k].copy_(v)
            graphed.sim._sets[0][k].copy_(v)
        actions = torch.rand(n, eager.num_actions, generator=gen, device=DEV) * 2 - 1
        if i == 2:  # flag a third of the environments so that the in-kernel reset draws are exercised
            for t in (eager, graphed):
                t.reset_buf[::3] = 1
        eager._draws = warm * draws_per_step + (i - 1)  # the offsets replay i draws with
        want = eager.step(actions)
        got = gs.step(actions)
        torch.cuda.synchronize()
        assert graphed.control_steps == eager.control_steps
        assert torch.equal(got[0]["obs"], want[0]["obs"]), "obs, step %d" % i
        assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]), "rew / reset, step %d" % i
        assert torch.equal(got[3]["time_outs"], want[3]["time_outs"])
        assert torch.equal(graphed.progress_buf, eager.progress_buf)
        for k in eager.sim.tensors:
            a, b = eager.sim.tensors[k], graphed.sim.tensors[k]
            if torch.is_tensor(a) and a.shape == b.shape and k != "workspace":  # scratch: contents unspecified
                assert torch.equal(a, b), "%s, step %d" % (k, i)


def test_graphed_step_refuses_tasks_that_need_the_host():
    from isaacgym_amd import tasks
    from isaacgym_amd.graph import GraphedStep
    task = tasks.make("Ant", 64, DEV)
    task.randomize = True
    with pytest.raises(ValueError):
        GraphedStep(task)
    with pytest.raises(ValueError):
        GraphedStep(tasks.make("CartpoleSample", 64, DEV, sample=True))  # its recorder counts captures on the host


def test_humanoid_amp_reset_and_step_replay_from_a_graph(tmp_path):
    """HumanoidAMP: device-only resets (reset_flagged) + step in one recorded graph against an eager twin."""
    from isaacgym_amd import cfg as C
    from isaacgym_amd import tasks
    from isaacgym_amd.graph import GraphedStep
    from isaacgym_amd.sim import SyntheticSim
    from isaacgym_amd.tasks.amp.utils_amp.motion_lib import MotionLib
    from tests import helpers as H
    lib = MotionLib(H.write_golden_motions(H.load_golden("motion"), str(tmp_path)), 28, [5, 8, 11, 14], DEV)
    n, warm = 2000, 2

    def make():
        cfg = C.task_config("HumanoidAMP", n)
        t = tasks.HumanoidAMP(cfg, DEV, DEV, sim=SyntheticSim("HumanoidAMP", n, DEV, seed=7, num_amp_obs_steps=2), motion_lib=lib)
        t.sim._sets = t.sim._sets[:1]
        return t

    eager, graphed = make(), make()
    gs = GraphedStep(graphed, warmup=warm, prologue=graphed.reset_flagged)
    zero = torch.zeros(n, 28, device=DEV)
    for _ in range(warm):
        eager.reset_flagged()
        eager.step(zero)
    gen = torch.Generator(device=DEV).manual_seed(9)
    for i in range(1, 4):
        actions = torch.rand(n, 28, generator=gen, device=DEV) * 2 - 1
        if i == 2:
            for t in (eager, graphed):
                t.reset_buf[::4] = 1
        eager.reset_flagged()
        want = eager.step(actions)
        got = gs.step(actions)
        torch.cuda.synchronize()
        assert torch.equal(got[0]["obs"], want[0]["obs"]) and torch.equal(got[2], want[2]), i
        assert torch.equal(graphed._amp_obs_buf, eager._amp_obs_buf), i
        assert torch.equal(graphed._dof_state, eager._dof_state) and torch.equal(graphed.progress_buf, eager.progress_buf), i
