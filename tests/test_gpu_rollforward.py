"""Long roll-forwards: a kernel and the CPU checker each carry their own persistent state (flags, progress,
potentials, reset joints, AMP history) over hundreds of steps while fresh simulator tensors arrive every step.
Single-step parity cannot see state that goes stale between steps, an ordering slip that only shows when an
environment resets twice, or an episode counter that wraps; this does."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("humanoid", [False, True])
def test_loco_roll_forward(humanoid):
    spec = synthetic.HUMANOID if humanoid else synthetic.ANT
    step, ostep = (ops.humanoid_step, oracle.humanoid_step) if humanoid else (ops.ant_step, oracle.ant_step)
    n, steps = 1500, 250
    p = synthetic.loco_params(spec, seed=5, rng_offset=0)
    p.max_episode_length_m1 = 39.0  # every environment times out several times
    t = synthetic.loco_state(spec, n, 3, DEV)
    t["progress"] = torch.randint(0, 40, (n,), device=DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    cache = {}
    nobs = spec["num_obs"]
    nonangle = [c for c in range(nobs) if c not in H.LOCO_ANGLE_COLS]
    physics = ("root_states", "sensors", "actions") + (("dof_force",) if humanoid else ())
    resets = 0
    for s in range(steps):
        fresh = synthetic.loco_state(spec, n, 1000 + s, DEV)
        for k in physics:  # the simulator's tensors of this step, written in place as an engine does
            t[k].copy_(fresh[k])
            ref[k][...] = fresh[k].cpu().numpy()
        # joint state: the engine moves every joint that was not just reset; a reset joint keeps what the kernel wrote
        moved = (t["reset"] == 0).view(n, 1, 1)
        t["dof_state"].copy_(torch.where(moved, fresh["dof_state"], t["dof_state"]))
        ref["dof_state"][...] = t["dof_state"].cpu().numpy()
        p.rng_offset = s
        resets += int(t["reset"].sum())
        step(p, t, n, cache)
        ostep(p, ref, n)
        what = "step %d" % s
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        for k in ("dof_state", "potentials", "prev_potentials", "rew"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)
        if s % 10 == 0 or s == steps - 1:
            obs = t["obs"].cpu().numpy()
            assert H.bit_equal_fraction(obs[:, nonangle], ref["obs"][:, nonangle]) == 1.0, what
            H.assert_angle_close(obs[:, list(H.LOCO_ANGLE_COLS)], ref["obs"][:, list(H.LOCO_ANGLE_COLS)], what=what,
                                 branch_cut=H.loco_branch_cut(humanoid))
    assert resets > 5 * n, resets  # every environment went through several episodes
    assert int(t["randomize"].min()) == steps


def test_cartpole_roll_forward():
    n, steps = 3000, 400
    p = synthetic.cartpole_params(seed=9, rng_offset=0)
    p.max_episode_length_m1 = 29.0
    t = synthetic.cartpole_state(n, 1, DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    cache = {}
    for s in range(steps):
        fresh = synthetic.cartpole_state(n, 500 + s, DEV)
        moved = (t["reset"] == 0).view(n, 1, 1)
        t["dof_state"].copy_(torch.where(moved, fresh["dof_state"].view(n, 2, 2), t["dof_state"].view(n, 2, 2)).view_as(t["dof_state"]))
        ref["dof_state"][...] = t["dof_state"].cpu().numpy()
        p.rng_offset = s
        ops.cartpole_step(p, t, n, cache)
        oracle.cartpole_step(p, ref, n)
        for k in ("reset", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, s)
        for k in ("dof_state", "obs", "rew"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, s)


@pytest.mark.parametrize("K", [2, 3])
def test_amp_roll_forward(K):
    """The AMP history: after every step slot k of every environment holds the observation of k steps ago (the
    kernel's and the checker's own), also across the early terminations and time-outs of 200 steps."""
    n, steps = 700, 200
    p = synthetic.amp_params(K, False, True, 25)
    t = synthetic.amp_state(n, 2, DEV, K, 25)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    cache = {}
    history = []
    for s in range(steps):
        fresh = synthetic.amp_state(n, 700 + s, DEV, K, 25)
        for k in ("root_states", "dof_state", "rigid_body_state", "contact_forces"):
            t[k].copy_(fresh[k])
            ref[k][...] = fresh[k].cpu().numpy()
        # the learner's reset between steps: progress back to 0 where the step asked for it (reset_idx zeroes the flags)
        done = t["reset"] != 0
        t["progress"][done] = 0
        t["reset"][done] = 0
        ref["progress"][...] = t["progress"].cpu().numpy()
        ref["reset"][...] = t["reset"].cpu().numpy()
        ops.humanoid_amp_step(p, t, n, cache)
        oracle.humanoid_amp_step(p, ref, n)
        what = "step %d" % s
        for k in ("reset", "terminate", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        got = t["amp_obs"].cpu().numpy().reshape(n, K, 105)
        history.insert(0, got[:, 0].copy())
        for k in range(1, min(K, len(history))):  # slot k is the kernel's own observation of k steps ago, bit for bit
            assert np.array_equal(got[:, k].view(np.uint32), history[k].view(np.uint32)), (what, k)
        del history[K:]
        if s % 10 == 0 or s == steps - 1:
            H.assert_amp_obs_close(got[:, 0], ref["amp_obs"].reshape(n, K, 105)[:, 0], what=what, quiet=True)
            H.assert_amp_obs_close(t["obs"].cpu().numpy(), ref["obs"], what=what + " obs", quiet=True)
    assert int((t["progress"] == 0).sum()) < n  # and the episodes did run
