"""GPU parity of the AllegroHand pre/post physics kernels: replay of the
reference's golden recordings and comparison with the CPU checker on larger
seeded inputs (in-kernel Philox resets)."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("variant", H.ALLEGRO_VARIANTS)
def test_allegro_golden(variant):
    g = H.load_golden("allegro_" + variant)
    H.allegro_replay(g, ops.allegro_pre_physics, ops.allegro_post_physics,
                     to_buf=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV),
                     to_np=lambda t: t.cpu().numpy(),
                     zeros=lambda shape, dt: torch.zeros(shape, device=DEV,
                                                         dtype=torch.int64 if dt == np.int64 else torch.float32),
                     force_fn=ops.hand_random_forces)


@pytest.mark.parametrize("n,obs_type,asym", [(0, "full_state", False), (1, "full_state", False),
                                             (67, "full", True), (1000, "full_no_vel", True),
                                             (300007, "full_state", True), (1 << 20, "full_state", False)])
def test_allegro_vs_oracle(n, obs_type, asym):
    p = synthetic.allegro_params(obs_type, seed=99, rng_offset=3)
    t = synthetic.allegro_state(n, 21, DEV, obs_type, asym)
    # the tail of VecTask.step folded into the post launch: clamped copies (a clip that bites) and time-out flags
    p.clip_obs = 0.5
    t["obs_clamped"] = torch.full_like(t["obs"], 7.0)
    t["states_clamped"] = torch.full((n, 88), 7.0, device=DEV) if asym else None
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in t.items()}
    for step in range(2):
        p.rng_offset = 3 + step
        ops.allegro_pre_physics(p, t, n)
        torch.cuda.synchronize()
        if n == 0:
            ops.allegro_post_physics(p, t, n)
            continue
        oracle.allegro_pre_physics(p, ref, n)
        # reset rotations come out of sin/cos (math-library dependent, 1e-5 class)
        H.assert_close(t["goal_states"].cpu().numpy(), ref["goal_states"], what="goal_states")
        H.assert_close(t["root_state_tensor"].cpu().numpy(), ref["root_state_tensor"], what="root rows")
        # the reward is ill-conditioned in the rotation inputs where |quat_diff.xyz| -> 1
        # (asin slope -> inf), so both sides continue from the same rotations
        t["goal_states"].copy_(torch.from_numpy(ref["goal_states"]))
        t["root_state_tensor"].copy_(torch.from_numpy(ref["root_state_tensor"]))
        ops.allegro_post_physics(p, t, n)
        torch.cuda.synchronize()
        oracle.allegro_post_physics(p, ref, n)
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


def test_allegro_two_shards_equal_one_on_device():
    """Two environment shards + gathered records + finalize == the unsharded call."""
    from isaacgym_amd import distributed
    n = 50000
    p = synthetic.allegro_params("full_state", seed=1, rng_offset=2)
    full = synthetic.allegro_state(n, 8, DEV, "full_state")
    full["rand"] = torch.rand(n, 45, device=DEV)
    full["consecutive_successes"].fill_(0.25)
    shards = []
    for r in range(2):
        s, c = distributed.env_shard(n, r, 2)
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
    ops.allegro_post_physics(p, full, n)
    for _, c, sh in shards:
        ops.allegro_post_physics_shard(p, sh, c)
    records = torch.cat([sh["workspace"].view(torch.uint8)[:16] for _, _, sh in shards]).contiguous()
    assert int(records.view(torch.int64)[0]) + int(records.view(torch.int64)[2]) == int(full["reset"].sum())
    for s, c, sh in shards:
        ops.allegro_finalize_shards(p, sh, records)
        for k in ("obs", "rew", "reset", "reset_goal", "progress", "successes"):
            assert torch.equal(sh[k], full[k][s:s + c]), k
        H.assert_close(sh["consecutive_successes"].cpu().numpy(), full["consecutive_successes"].cpu().numpy(),
                       rtol=1e-6, what="consecutive_successes")
    assert torch.equal(shards[0][2]["consecutive_successes"], shards[1][2]["consecutive_successes"])


@pytest.mark.parametrize("pattern", ["stride1", "stride2", "stride4", "stride5", "permuted", "mixed_tiles", "last_rows"])
@pytest.mark.parametrize("obs_type", ["full_state", "full_no_vel"])
def test_allegro_object_index_patterns(pattern, obs_type):
    """Evenly spaced actor indices take the streamed-window path, anything else the row gather;
    both must give the bits of the standard (3 * env + 1) layout.  'last_rows' puts the window
    flush against the end of the tensor (no read may go past it: the allocation ends there)."""
    n = 64 * 5 + 17
    p = synthetic.allegro_params(obs_type, seed=5, rng_offset=1)
    base = synthetic.allegro_state(n, 33, DEV, obs_type, False)
    rows = base["root_state_tensor"][base["object_indices"]].clone()
    env = torch.arange(n, device=DEV)
    g = torch.Generator().manual_seed(7)
    if pattern.startswith("stride"):
        s = int(pattern[6:])
        idx, total = s * env + (s - 1), s * n
    elif pattern == "permuted":
        idx, total = torch.randperm(3 * n, generator=g)[:n].to(DEV), 3 * n
    elif pattern == "mixed_tiles":
        idx = 3 * env + 1
        tile = idx[64:128].clone()
        idx[64:128] = tile[torch.randperm(64, generator=g).to(DEV)]
        idx[200] += 1  # one irregular index inside an otherwise regular tile
        total = 3 * n + 1
    else:  # last_rows: stride 3 with the object row last in its env, tensor ends right after
        idx, total = 3 * env + 2, 3 * n
    root = torch.randn(total, 13, device=DEV)
    root[idx] = rows
    t = {k: (v.clone() if v is not None else None) for k, v in base.items()}
    t["root_state_tensor"], t["object_indices"] = root.contiguous(), idx.contiguous()
    ops.allegro_post_physics(p, base, n)
    ops.allegro_post_physics(p, t, n)
    for k in ("obs", "rew", "reset", "reset_goal", "progress", "successes", "consecutive_successes"):
        assert torch.equal(base[k].view(torch.uint8), t[k].view(torch.uint8)), k
