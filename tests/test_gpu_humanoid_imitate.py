"""HumanoidImitate on the GPU: recording of the reference class, comparison with the CPU
checker on larger inputs, and the task class."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic, tasks
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("recording", ["humanoid_imitate", "humanoid_imitate_extreme"])
def test_humanoid_imitate_golden(recording):
    H.him_replay(H.load_golden(recording), ops.humanoid_imitate_step,
                 to_buf=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV), to_np=lambda t: t.cpu().numpy(),
                 zeros=lambda shape, dt: torch.zeros(shape, device=DEV,
                                                     dtype=torch.int64 if dt == np.int64 else torch.float32))


@pytest.mark.parametrize("n,with_rand", [(0, False), (1, True), (15, False), (16, True), (1000, False),
                                         (50007, True), (1 << 18, False)])
def test_humanoid_imitate_vs_oracle(n, with_rand):
    p = synthetic.him_params(seed=7, rng_offset=3)
    t = synthetic.him_state(n, 5, DEV, with_rand)
    # the tail of VecTask.step folded into the launch: clamped copy (a clip that bites) and time-out flags
    p.clip_obs = 0.6
    t["obs_clamped"] = torch.full((n, 205), 7.0, device=DEV)
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    for step in range(2):
        p.rng_offset = 3 + step
        ops.humanoid_imitate_step(p, t, n)
        torch.cuda.synchronize()
        if n == 0:
            continue
        oracle.humanoid_imitate_step(p, ref, n)
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), "%s step %d" % (k, step)
        for k in ("obs", "dof_state", "potentials", "prev_potentials"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, "%s step %d" % (k, step)
        H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what="rew step %d" % step)
        assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), "time_outs step %d" % step
        assert np.array_equal(ref["time_outs"], (ref["progress"].astype(np.float32) >= p.max_episode_length_m1) & (ref["reset"] != 0))
        assert torch.equal(t["obs_clamped"], torch.clamp(t["obs"], -0.6, 0.6)), "obs_clamped step %d" % step
        assert np.array_equal(ref["obs_clamped"], np.clip(ref["obs"], -0.6, 0.6))


def test_humanoid_imitate_class():
    n = 3000
    task = tasks.make("HumanoidImitate", n, DEV, seed=1)
    assert task.obs_buf.shape == (n, 205) and task.num_actions == 63
    task.reset_done()  # raises the flags; the first step applies them
    obs_dict, rew, reset, extras = task.step(torch.zeros(n, 63, device=DEV))
    assert int(task.progress_buf.max()) == 0 and obs_dict["obs"].shape == (n, 205)
    lo, hi = task.dof_limits_lower, task.dof_limits_upper
    assert bool(((task.dof_pos >= lo) & (task.dof_pos <= hi)).all())
    assert torch.equal(task.obs_buf[:, 126:139], task.root_states)
    assert torch.equal(task.obs_buf[:, 139:205].view(n, 22, 3), task.rigid_body_states.view(n, 22, 13)[..., 0:3])
    assert bool((rew > 0).all()) and extras["time_outs"].dtype == torch.bool
