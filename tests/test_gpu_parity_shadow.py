"""GPU parity of the ShadowHand pre/post physics kernels: replay of the reference's golden
recordings and comparison with the CPU checker on larger seeded inputs (in-kernel Philox resets)."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("variant", H.SHADOW_VARIANTS)
def test_shadow_golden(variant):
    g = H.load_golden("shadow_" + variant)
    H.shadow_replay(g, ops.shadow_pre_physics, ops.shadow_post_physics,
                    to_buf=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV),
                    to_np=lambda t: t.cpu().numpy(),
                    zeros=lambda shape, dt: torch.zeros(shape, device=DEV,
                                                        dtype=torch.int64 if dt == np.int64 else torch.float32),
                     force_fn=ops.hand_random_forces)


@pytest.mark.parametrize("n,obs_type,asym", [(0, "full_state", False), (1, "full_state", False), (31, "openai", False),
                                             (67, "full", True), (1000, "full_no_vel", True), (33, "openai", True),
                                             (100003, "full_state", True), (1 << 18, "full_state", False)])
def test_shadow_vs_oracle(n, obs_type, asym):
    p = synthetic.shadow_params(obs_type, seed=99, rng_offset=3)
    t = synthetic.shadow_state(n, 21, DEV, obs_type, asym)
    # the tail of VecTask.step folded into the post launch: clamped copies (a clip that bites) and time-out flags
    p.clip_obs = 0.5
    t["obs_clamped"] = torch.full_like(t["obs"], 7.0)
    t["states_clamped"] = torch.full((n, 211), 7.0, device=DEV) if asym else None
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in t.items()}
    for step in range(2):
        p.rng_offset = 3 + step
        ops.shadow_pre_physics(p, t, n)
        torch.cuda.synchronize()
        if n == 0:
            ops.shadow_post_physics(p, t, n)
            continue
        oracle.shadow_pre_physics(p, ref, n)
        # reset rotations come out of sin/cos (math-library dependent, 1e-5 class)
        H.assert_close(t["goal_states"].cpu().numpy(), ref["goal_states"], what="goal_states")
        H.assert_close(t["root_state_tensor"].cpu().numpy(), ref["root_state_tensor"], what="root rows")
        # the reward is ill-conditioned in the rotation inputs where |quat_diff.xyz| -> 1, so both
        # sides continue from the same rotations
        t["goal_states"].copy_(torch.from_numpy(ref["goal_states"]))
        t["root_state_tensor"].copy_(torch.from_numpy(ref["root_state_tensor"]))
        ops.shadow_post_physics(p, t, n)
        torch.cuda.synchronize()
        oracle.shadow_post_physics(p, ref, n)
        for k in ("reset", "reset_goal", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), "%s step %d" % (k, step)
        for k in ("successes", "prev_targets", "cur_targets", "dof_state"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, "%s step %d" % (k, step)
        assert H.bit_equal_fraction(t["obs"].cpu().numpy(), ref["obs"]) == 1.0, "obs step %d" % step
        if asym:
            assert H.bit_equal_fraction(t["states"].cpu().numpy(), ref["states"]) == 1.0, "states step %d" % step
        H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what="rew step %d" % step)
        H.assert_close(t["consecutive_successes"].cpu().numpy(), ref["consecutive_successes"], what="cons")
        assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), "time_outs step %d" % step
        assert np.array_equal(ref["time_outs"], (ref["progress"] >= 599) & (ref["reset"] != 0))
        assert torch.equal(t["obs_clamped"], torch.clamp(t["obs"], -0.5, 0.5)), "obs_clamped step %d" % step
        assert np.array_equal(ref["obs_clamped"], np.clip(ref["obs"], -0.5, 0.5))
        if asym:
            assert torch.equal(t["states_clamped"], torch.clamp(t["states"], -0.5, 0.5)), "states_clamped step %d" % step


def test_shadow_two_shards_equal_one_on_device():
    """Two environment shards + rank-ordered records reproduce the one-launch result bit for bit."""
    n, cut = 3000, 1216
    p = synthetic.shadow_params("full_state", seed=3)
    full = synthetic.shadow_state(n, 8, DEV, "full_state", False)
    shards = []
    for s, c in ((0, cut), (cut, n - cut)):
        sh = {}
        for k, v in full.items():
            if v is None:
                sh[k] = None
            elif k == "root_state_tensor":
                sh[k] = v[3 * s:3 * (s + c)].clone()
            elif k in ("object_indices", "goal_object_indices"):
                sh[k] = (v[s:s + c] - 3 * s).contiguous()
            elif k == "workspace":
                sh[k] = torch.full_like(v, float("nan"))  # scratch needs no initialisation
            elif k == "consecutive_successes":
                sh[k] = v.clone()
            else:
                sh[k] = v[s:s + c].clone()
        shards.append((s, c, sh))
    ops.shadow_post_physics(p, full, n)
    for _, c, sh in shards:
        ops.shadow_post_physics_shard(p, sh, c)
    records = torch.cat([sh["workspace"].view(torch.uint8)[:16] for _, _, sh in shards]).contiguous()
    assert int(records.view(torch.int64)[0]) + int(records.view(torch.int64)[2]) == int(full["reset"].sum())
    for s, c, sh in shards:
        ops.shadow_finalize_shards(p, sh, records)
        for k in ("obs", "rew", "reset", "reset_goal", "progress", "successes"):
            assert torch.equal(sh[k].view(torch.uint8), full[k][s:s + c].view(torch.uint8)), k
        assert torch.equal(sh["consecutive_successes"].view(torch.uint8), full["consecutive_successes"].view(torch.uint8))


def test_shadow_rejects_bad_handles():
    p = synthetic.shadow_params("full_state", fingertip_handles=[7, 11, 15, 20, 28])
    t = synthetic.shadow_state(64, 1, DEV)
    with pytest.raises(RuntimeError):
        ops.shadow_post_physics(p, t, 64)
    p = synthetic.shadow_params("full_state", actuated_dof_indices=[0] * 20)
    with pytest.raises(RuntimeError):
        ops.shadow_pre_physics(p, t, 64)
