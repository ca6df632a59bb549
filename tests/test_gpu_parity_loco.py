"""GPU parity: HIP kernels (through the C ABI) against the golden vectors and
against the CPU checker on seeded synthetic inputs.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", ["cartpole", "cartpole_sample", "cartpole_nonfinite", "cartpole_extreme"])
def test_cartpole_golden(name):
    g = H.load_golden(name)
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = H.cartpole_params(m)
    reset, progress = _to_dev(g["reset0"]), _to_dev(g["progress0"])
    for t in range(T):
        ds = _to_dev(g["in_dof_state"][t])
        obs = torch.zeros(N, 4, device=DEV)
        rew = torch.zeros(N, device=DEV)
        ops.cartpole_step(p, dict(dof_state=ds, rand=_to_dev(g["rand"][t]), obs=obs, rew=rew, reset=reset,
                                  progress=progress), N)
        assert np.array_equal(reset.cpu().numpy(), g["reset"][t])
        assert np.array_equal(progress.cpu().numpy(), g["progress"][t])
        assert H.bit_equal_fraction(ds.cpu().numpy(), g["out_dof_state"][t]) == 1.0
        assert H.bit_equal_fraction(obs.cpu().numpy(), g["obs"][t]) == 1.0
        assert H.bit_equal_fraction(rew.cpu().numpy(), g["rew"][t]) == 1.0


@pytest.mark.parametrize("n,with_rand", [(0, False), (1, True), (255, True), (4096, False), (1000003, False)])
def test_cartpole_vs_oracle(n, with_rand):
    p = synthetic.cartpole_params(seed=77, rng_offset=5)
    t = synthetic.cartpole_state(n, 3, DEV, with_rand)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    ops.cartpole_step(p, t, n)
    torch.cuda.synchronize()
    if n == 0:
        return
    oracle.cartpole_step(p, ref, n)
    for k in ("reset", "progress"):
        assert np.array_equal(t[k].cpu().numpy(), ref[k]), k
    for k in ("dof_state", "obs", "rew"):
        assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, k


def test_cartpole_writes_the_step_tail_in_the_same_launch():
    """obs_clamped / time_outs (the tail of VecTask.step) out of the step kernel, against the CPU restatement."""
    n = 70001
    p = synthetic.cartpole_params(seed=5, rng_offset=1)
    p.clip_obs = 0.8
    t = synthetic.cartpole_state(n, 4, DEV, True)
    t["progress"][::5] = 498
    t["obs_clamped"] = torch.full((n, 4), 9.0, device=DEV)
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    ops.cartpole_step(p, t, n)
    oracle.cartpole_step(p, ref, n)
    for k in ("reset", "progress", "time_outs"):
        assert np.array_equal(t[k].cpu().numpy(), ref[k]), k
    assert 0 < int(ref["time_outs"].sum()) < n
    assert np.array_equal(ref["time_outs"], (ref["progress"] >= 499) & (ref["reset"] != 0))
    assert torch.equal(t["obs_clamped"], torch.clamp(t["obs"], -0.8, 0.8))
    assert np.array_equal(ref["obs_clamped"], np.clip(ref["obs"], -0.8, 0.8))
    t["obs_clamped"] = t["obs"]
    with pytest.raises(RuntimeError, match="alias"):
        ops.cartpole_step(p, t, n)


def _golden_loco(name, step, nobs, humanoid):
    g = H.load_golden(name)
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = H.loco_params(m, g)
    reset, progress, pot = _to_dev(g["reset0"]), _to_dev(g["progress0"]), _to_dev(g["potentials0"])
    randomize = torch.zeros(N, dtype=torch.int64, device=DEV)
    init_root, init_dof = _to_dev(g["initial_root_states"]), _to_dev(g["initial_dof_pos"])
    nonangle = [c for c in range(nobs) if c not in H.LOCO_ANGLE_COLS]
    ang = list(H.LOCO_ANGLE_COLS)
    for t in range(T):
        ds = _to_dev(g["in_dof_state"][t])
        out = dict(obs=torch.zeros(N, nobs, device=DEV), prev_potentials=torch.zeros(N, device=DEV),
                   up_vec=torch.zeros(N, 3, device=DEV), heading_vec=torch.zeros(N, 3, device=DEV),
                   rew=torch.zeros(N, device=DEV))
        bufs = dict(root_states=_to_dev(g["in_root"][t]), dof_state=ds, sensors=_to_dev(g["in_sensors"][t]),
                    actions=_to_dev(g["in_actions"][t]), initial_root_states=init_root,
                    initial_dof_pos=init_dof, rand=_to_dev(g["rand"][t]), potentials=pot, reset=reset,
                    progress=progress, randomize=randomize, **out)
        if humanoid:
            bufs["dof_force"] = _to_dev(g["in_dof_force"][t])
        else:
            bufs["true_objective"] = torch.zeros(N, device=DEV)
        step(p, bufs, N)
        assert np.array_equal(reset.cpu().numpy(), g["reset"][t]), "reset step %d" % t
        assert np.array_equal(progress.cpu().numpy(), g["progress"][t]), "progress step %d" % t
        assert np.array_equal(randomize.cpu().numpy(), g["randomize"][t])
        obs = out["obs"].cpu().numpy()
        # tolerance bar of the task statement (1e-5 relative, natural scale) ...
        H.assert_close(obs[:, nonangle], g["obs"][t][:, nonangle], what="obs step %d" % t)
        H.assert_angle_close(obs[:, ang], g["obs"][t][:, ang], what="obs angles step %d" % t, branch_cut=H.loco_branch_cut(humanoid))
        H.assert_close(out["rew"].cpu().numpy(), g["rew"][t], what="rew step %d" % t)
        # ... and the stronger property the exact arithmetic model gives:
        for key, arr in (("out_dof_state", ds), ("potentials", pot), ("prev_potentials", out["prev_potentials"]),
                         ("up_vec", out["up_vec"]), ("heading_vec", out["heading_vec"]), ("rew", out["rew"])):
            f = H.bit_equal_fraction(arr.cpu().numpy(), g[key][t])
            assert f == 1.0, "%s step %d bit-equal fraction %f" % (key, t, f)
        assert H.bit_equal_fraction(obs[:, nonangle], g["obs"][t][:, nonangle]) == 1.0
        if not humanoid:
            assert H.bit_equal_fraction(bufs["true_objective"].cpu().numpy(), g["true_objective"][t]) == 1.0


def test_ant_golden():
    _golden_loco("ant", ops.ant_step, 60, False)


def test_humanoid_golden():
    _golden_loco("humanoid", ops.humanoid_step, 108, True)


def test_ant_golden_with_every_parameter_changed():
    _golden_loco("ant_alt", ops.ant_step, 60, False)


def test_humanoid_golden_with_every_parameter_changed():
    _golden_loco("humanoid_alt", ops.humanoid_step, 108, True)


def test_ant_golden_on_non_finite_state():
    """NaN / +-Inf in the simulator tensors of some environments: every output as the reference leaves it
    (NaN where it has NaN, flags exact), and no other environment touched."""
    _golden_loco("ant_nonfinite", ops.ant_step, 60, False)


def test_humanoid_golden_on_non_finite_state():
    _golden_loco("humanoid_nonfinite", ops.humanoid_step, 108, True)


def _vs_oracle(spec, step, ostep, n, with_rand, steps=2):
    p = synthetic.loco_params(spec, seed=1234, rng_offset=9)
    t = synthetic.loco_state(spec, n, 11, DEV, with_rand)
    if not spec["humanoid"]:
        t["true_objective"] = torch.zeros(n, device=DEV)
    # the tail of VecTask.step folded into the launch: clamped copy (a clip that bites) and time-out flags
    p.clip_obs = 0.75
    t["obs_clamped"] = torch.full((n, spec["num_obs"]), 7.0, device=DEV)
    t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    nobs = spec["num_obs"]
    nonangle = [c for c in range(nobs) if c not in H.LOCO_ANGLE_COLS]
    ang = list(H.LOCO_ANGLE_COLS)
    for s in range(steps):
        p.rng_offset = 9 + s
        step(p, t, n)
        torch.cuda.synchronize()
        if n == 0:
            continue
        ostep(p, ref, n)
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), "%s step %d" % (k, s)
        obs = t["obs"].cpu().numpy()
        assert H.bit_equal_fraction(obs[:, nonangle], ref["obs"][:, nonangle]) == 1.0, "obs step %d" % s
        H.assert_angle_close(obs[:, ang], ref["obs"][:, ang], what="angles step %d" % s, branch_cut=H.loco_branch_cut(spec["humanoid"]))
        for k in ("dof_state", "potentials", "prev_potentials", "up_vec", "heading_vec", "rew"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, "%s step %d" % (k, s)
        assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), "time_outs step %d" % s
        assert np.array_equal(ref["time_outs"], (ref["progress"] >= spec["max_episode_length"] - 1) & (ref["reset"] != 0))
        assert torch.equal(t["obs_clamped"], torch.clamp(t["obs"], -0.75, 0.75)), "obs_clamped step %d" % s
        assert float(t["obs_clamped"].abs().max()) == 0.75
        # next step: new "physics" state on both sides
        nxt = synthetic.loco_state(spec, n, 100 + s, DEV, with_rand)
        for k in ("root_states", "dof_state", "sensors", "actions") + (("dof_force",) if spec["humanoid"] else ()):
            t[k].copy_(nxt[k])
            ref[k][...] = nxt[k].cpu().numpy()


@pytest.mark.parametrize("n,with_rand", [(0, False), (1, True), (63, True), (257, False), (100003, True),
                                         (1 << 20, False)])
def test_ant_vs_oracle(n, with_rand):
    _vs_oracle(synthetic.ANT, ops.ant_step, oracle.ant_step, n, with_rand)


@pytest.mark.parametrize("n,with_rand", [(1, True), (129, False), (50001, True), (1 << 20, False)])
def test_humanoid_vs_oracle(n, with_rand):
    _vs_oracle(synthetic.HUMANOID, ops.humanoid_step, oracle.humanoid_step, n, with_rand)


def test_rejects_bad_arguments():
    from isaacgym_amd._lib import IgxError
    p = synthetic.loco_params(synthetic.ANT)
    t = synthetic.loco_state(synthetic.ANT, 64, 0, DEV)
    bad = dict(t)
    bad["obs"] = None
    with pytest.raises(IgxError):
        ops.ant_step(p, bad, 64)
    with pytest.raises(IgxError):
        ops.humanoid_step(p, t, 64)  # wrong num_dof for this entry point
    with pytest.raises(ValueError):
        ops.ant_step(p, dict(t, obs=t["obs"].cpu()), 64)
    with pytest.raises(ValueError):
        ops.ant_step(p, dict(t, dof_state=t["dof_state"][..., 0]), 64)  # strided view


@pytest.mark.parametrize("spec,reset", [(synthetic.ANT, ops.ant_reset), (synthetic.HUMANOID, ops.humanoid_reset)])
@pytest.mark.parametrize("m_frac,with_rand", [(0.0, False), (0.01, True), (1.0, False)])
def test_loco_reset_idx_vs_oracle(spec, reset, m_frac, with_rand):
    n = 70001
    p = synthetic.loco_params(spec, seed=42, rng_offset=17)
    t = synthetic.loco_state(spec, n, 6, DEV, with_rand)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    rng = np.random.default_rng(0)
    ids = np.sort(rng.choice(n, size=int(n * m_frac), replace=False)).astype(np.int64)
    reset(p, t, _to_dev(ids))
    torch.cuda.synchronize()
    oracle.loco_reset(p, ref, ids)
    for k in ("dof_state", "potentials", "prev_potentials", "progress", "reset", "root_states", "obs"):
        assert np.array_equal(t[k].cpu().numpy().view(np.uint8), ref[k].view(np.uint8)), k


@pytest.mark.parametrize("with_rand", [False, True])
def test_cartpole_reset_idx_vs_oracle(with_rand):
    n = 4096
    p = synthetic.cartpole_params(seed=3, rng_offset=4)
    t = synthetic.cartpole_state(n, 2, DEV, with_rand)
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    ids = np.arange(0, n, 3).astype(np.int64)
    ops.cartpole_reset(p, t, _to_dev(ids))
    oracle.cartpole_reset(p, ref, ids)
    for k in ("dof_state", "progress", "reset"):
        assert np.array_equal(t[k].cpu().numpy().view(np.uint8), ref[k].view(np.uint8)), k
    ops.cartpole_reset(p, t, _to_dev(np.zeros(0, np.int64)))  # empty list is a no-op


@pytest.mark.parametrize("spec_name", ["ant", "humanoid"])
def test_non_finite_inputs_stay_local(spec_name):
    """NaN / inf in the simulator's tensors must neither trap nor leak into other environments:
    the clean environments keep the bits of a run without the poison, and the poisoned ones get
    the flags the reference's comparisons give (every comparison with NaN is false)."""
    spec = synthetic.ANT if spec_name == "ant" else synthetic.HUMANOID
    step, ostep = (ops.ant_step, oracle.ant_step) if spec_name == "ant" else (ops.humanoid_step, oracle.humanoid_step)
    n = 64 * 7 + 5
    p = synthetic.loco_params(spec, seed=5, rng_offset=2)
    clean = synthetic.loco_state(spec, n, 3, DEV, True)
    if spec_name == "ant":
        clean["true_objective"] = torch.zeros(n, device=DEV)
    dirty = {k: v.clone() for k, v in clean.items()}
    bad = torch.tensor([0, 1, 2, 3, 64, 65, 130, 191, n - 1], device=DEV)
    dirty["root_states"][bad[0], 2] = float("nan")       # height: z < termination is false
    dirty["root_states"][bad[1], 3:7] = float("nan")      # orientation
    dirty["root_states"][bad[2], 0] = float("inf")
    dirty["root_states"][bad[3], 7:13] = float("-inf")
    dirty["dof_state"][bad[4]] = float("nan")
    dirty["sensors"][bad[5]] = float("inf")
    dirty["actions"][bad[6]] = float("nan")
    dirty["potentials"][bad[7]] = float("nan")
    dirty["root_states"][bad[8]] = float("nan")
    ref = {k: v.cpu().numpy().copy() for k, v in dirty.items()}
    step(p, clean, n)
    step(p, dirty, n)
    torch.cuda.synchronize()
    ostep(p, ref, n)
    ok = torch.ones(n, dtype=torch.bool, device=DEV)
    ok[bad] = False
    for k in ("obs", "rew", "reset", "progress", "potentials", "prev_potentials", "up_vec", "heading_vec", "dof_state"):
        assert torch.equal(dirty[k][ok].view(torch.uint8), clean[k][ok].view(torch.uint8)), k
    for k in ("reset", "progress", "randomize"):
        assert np.array_equal(dirty[k].cpu().numpy(), ref[k]), k
    # the poisoned rows agree with the CPU checker wherever it is finite, and are non-finite where it is not
    got, want = dirty["obs"].cpu().numpy(), ref["obs"]
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    nonangle = [c for c in range(spec["num_obs"]) if c not in H.LOCO_ANGLE_COLS]
    assert np.array_equal(got[:, nonangle][fin[:, nonangle]].view(np.uint32), want[:, nonangle][fin[:, nonangle]].view(np.uint32))
    assert np.array_equal(np.isnan(dirty["rew"].cpu().numpy()), np.isnan(ref["rew"]))


def test_device_resident_draw_counter_adds_to_rng_offset():
    """rng_offset_dev: the kernel draws with rng_offset + *counter (what a replayed graph relies on)."""
    n = 5000
    spec = synthetic.ANT
    t = synthetic.loco_state(spec, n, 21, DEV, with_rand=False)
    t["reset"][::2] = 1
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    counter = torch.tensor([7], dtype=torch.int64, device=DEV)
    p = synthetic.loco_params(spec, seed=99, rng_offset=5)
    p.rng_offset_dev = counter.data_ptr()
    ops.ant_step(p, t, n)
    torch.cuda.synchronize()
    oracle.ant_step(synthetic.loco_params(spec, seed=99, rng_offset=12), ref, n)
    for k in ("dof_state", "rew", "potentials"):
        assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, k
    nonangle = [c for c in range(60) if c not in H.LOCO_ANGLE_COLS]
    assert H.bit_equal_fraction(t["obs"].cpu().numpy()[:, nonangle], ref["obs"][:, nonangle]) == 1.0
    assert np.array_equal(t["reset"].cpu().numpy(), ref["reset"])
    # and the counter matters: without it the draws are those of rng_offset alone
    other = synthetic.loco_state(spec, n, 21, DEV, with_rand=False)
    other["reset"][::2] = 1
    ops.ant_step(synthetic.loco_params(spec, seed=99, rng_offset=5), other, n)
    assert not torch.equal(other["dof_state"], t["dof_state"])


@pytest.mark.parametrize("spec,step", [(synthetic.ANT, ops.ant_step), (synthetic.HUMANOID, ops.humanoid_step)])
def test_two_shards_draw_what_one_gpu_draws(spec, step):
    """In-kernel resets (rand == NULL): two environment shards that share the seed and pass their env_offset
    equal the unsharded call bit for bit."""
    from isaacgym_amd import distributed
    n = 10007
    full = synthetic.loco_state(spec, n, 3, DEV)
    full["reset"].fill_(0)
    full["reset"][::3] = 1
    shards = []
    for r in range(2):
        s, c = distributed.env_shard(n, r, 2)
        shards.append((s, c, {k: (v[s:s + c].clone() if v is not None else None) for k, v in full.items()}))
    p = synthetic.loco_params(spec, seed=77, rng_offset=5)
    step(p, full, n)
    for s, c, sh in shards:
        ps = synthetic.loco_params(spec, seed=77, rng_offset=5)
        ps.env_offset = s
        step(ps, sh, c)
        for k in ("dof_state", "obs", "rew", "reset", "progress", "potentials"):
            assert torch.equal(sh[k].view(torch.uint8), full[k][s:s + c].view(torch.uint8)), k
    s, c, sh = shards[1]
    again = synthetic.loco_state(spec, n, 3, DEV)
    again["reset"].fill_(0)
    again["reset"][::3] = 1
    wrong = {k: (v[s:s + c].clone() if v is not None else None) for k, v in again.items()}
    step(synthetic.loco_params(spec, seed=77, rng_offset=5), wrong, c)  # env_offset forgotten: other draws
    assert not torch.equal(wrong["dof_state"], full["dof_state"][s:s + c])


def test_cached_call_follows_a_repointed_tensor():
    """The per-task argument cache is keyed on the tensor objects AND on where they point: a tensor whose storage
    is swapped in place (`t.data = ...`) keeps its identity, and the cached struct must not launch on the old
    address (stale data) or skip the size check (out-of-bounds access)."""
    n = 1000
    spec = synthetic.ANT
    p = synthetic.loco_params(spec, seed=2, rng_offset=4)
    t = synthetic.loco_state(spec, n, 21, DEV)
    cache = {}
    ops.ant_step(p, t, n, cache)
    other = synthetic.loco_state(spec, n, 22, DEV)
    start = {k: v.clone() for k, v in t.items()}
    start["root_states"] = other["root_states"].clone()
    t["root_states"].data = other["root_states"]      # same object, other memory
    ops.ant_step(p, t, n, cache)
    ops.ant_step(p, start, n)                          # uncached call from the same state
    torch.cuda.synchronize()
    for k in ("obs", "rew", "potentials", "reset", "progress"):
        assert torch.equal(t[k], start[k]), k
    t["root_states"].data = torch.zeros(n // 2, 13, device=DEV)   # new memory, too short for n environments
    with pytest.raises(ValueError, match="root_states"):
        ops.ant_step(p, t, n, cache)


@pytest.mark.parametrize("name,nobs,humanoid", [("ant_extreme", 60, False), ("humanoid_extreme", 108, True),
                                                ("ant_nonunit", 60, False), ("humanoid_nonunit", 108, True)])
def test_loco_golden_on_extreme_values(name, nobs, humanoid):
    """The reference's recording with +-3e38, +-1e30, 1e-30, subnormals, signed zeros, NaN and +-Inf scattered over
    2 % of the simulator state and the actions: overflow, underflow, subnormal arithmetic and the sign of zero on
    the GPU as in torch on the CPU."""
    _golden_loco(name, ops.humanoid_step if humanoid else ops.ant_step, nobs, humanoid)
