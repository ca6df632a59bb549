"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py [task ...]

For every task the reference's task class is instantiated without its
constructor (PhysX is absent), given synthetic state tensors, and its own
post_physics_step / pre_physics_step / reset_idx methods are executed on CPU.
Inputs, the uniforms torch drew inside reset_idx, and all resulting buffers
are stored as small .npz files.  The fixtures hold data only - no reference
source text.  torch version and parameters are recorded in each file.
"""
import json
import math
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim  # noqa: E402

OUT = os.environ.get("IGX_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CFG = os.path.join(ref_shim.IGE, "cfg", "task")


class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


def _cfg(name):
    with open(os.path.join(CFG, name + ".yaml")) as f:
        return yaml.safe_load(f)


def _meta(**kw):
    kw["torch"] = torch.__version__
    kw["cpu_capability"] = torch.backends.cpu.get_cpu_capability()
    return json.dumps(kw)


def _rand_quat(g, n):
    """half random unit quaternions, half small perturbations of identity
    (roll / yaw near 0 exercise the remainder wrap)."""
    q = torch.randn(n, 4, generator=g)
    small = torch.randn(n, 4, generator=g) * 1e-3
    small[:, 3] = 1.0
    pick = torch.rand(n, generator=g) < 0.5
    q = torch.where(pick.unsqueeze(-1), q, small)
    # a few exact identities and exact axis flips
    q[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    q[1] = torch.tensor([0.0, 0.0, 1.0, 0.0])
    q[2] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    return q / q.norm(dim=-1, keepdim=True)


def _install_gym_stubs():
    ref_shim.install()
    gt = sys.modules["isaacgym.gymtorch"]
    gt.unwrap_tensor = lambda t: t
    gt.wrap_tensor = lambda t: t
    sys.modules["isaacgym.gymapi"].LOCAL_SPACE = 0  # an enum of the engine, only ever passed back to it


def _replay_draws(state, specs):
    """Re-draw what torch.rand / torch.randn produced after RNG state `state`; specs are
    ("rand" | "randn", shape) in call order, or a callable(previous draws) -> such a pair."""
    keep = torch.get_rng_state()
    torch.set_rng_state(state)
    out = []
    for spec in specs:
        kind, shape = spec(out) if callable(spec) else spec
        out.append(torch.rand(*shape) if kind == "rand" else torch.randn(*shape))
    torch.set_rng_state(keep)
    return out


def _force_draws(t, N, st, shapes, env_ids):
    """The draws of the random-force block of pre_physics_step (tasks/allegro_hand.py:558-559, :617-625),
    scattered to their environments: [N,5] = (u of the probability re-draw, Bernoulli u, three normals)."""
    specs = [("rand", s) for s in shapes] + [("rand", (N,)),
             lambda prev: ("randn", (int((prev[-1] < t.random_force_prob).sum()), 1, 3))]
    draws = _replay_draws(st, specs)
    dense = np.zeros((N, 5), np.float32)
    if len(env_ids) > 0:
        dense[env_ids.numpy(), 0] = draws[len(shapes) - 1].numpy()  # the (len(env_ids),) draw closes reset_idx's list
    dense[:, 1] = draws[-2].numpy()
    sel = (draws[-2] < t.random_force_prob).nonzero().squeeze(-1).numpy()
    dense[sel, 2:5] = draws[-1].numpy().reshape(-1, 3)
    return dense, len(sel)


def _replay_uniforms(state, shapes):
    """Re-draw what torch.rand produced after RNG state `state`."""
    keep = torch.get_rng_state()
    torch.set_rng_state(state)
    out = [torch.rand(*s) for s in shapes]
    torch.set_rng_state(keep)
    return out


# --------------------------------------------------------------------------
def gen_cartpole(seed=0, N=4096, T=4, name="cartpole", poison=False):
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.cartpole")
    cfg = _cfg("Cartpole")

    class Fake(mod.Cartpole):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    t.cfg = cfg
    t.reset_dist = cfg["env"]["resetDist"]
    t.max_episode_length = 500
    t.num_envs, t.num_dof, t.device = N, 2, "cpu"
    t.gym, t.sim = _Any(), None
    t.dof_state = torch.zeros(N * 2, 2)
    t.dof_pos = t.dof_state.view(N, 2, 2)[..., 0]
    t.dof_vel = t.dof_state.view(N, 2, 2)[..., 1]
    t.obs_buf = torch.zeros(N, 4)
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.ones(N, dtype=torch.long)
    t.progress_buf = torch.randint(0, 500, (N,), generator=g)
    t.progress_buf[:8] = torch.tensor([496, 497, 498, 499, 500, 0, 1, 2])

    rec = {k: [] for k in ("in_dof_state", "rand", "out_dof_state", "obs", "rew", "reset", "progress")}
    init = {"reset0": t.reset_buf.numpy().copy(), "progress0": t.progress_buf.numpy().copy()}
    torch.manual_seed(seed + 100)
    for step in range(T):
        ds = torch.empty(N, 2, 2)
        ds[:, 0, 0] = (torch.rand(N, generator=g) * 2 - 1) * 3.5
        ds[:, 1, 0] = (torch.rand(N, generator=g) * 2 - 1) * 2.0
        ds[:, :, 1] = torch.randn(N, 2, generator=g)
        ds[0, 0, 0] = 3.0       # exactly resetDist: not > 3.0
        ds[1, 1, 0] = math.pi / 2  # float32(pi/2): not > float32(pi/2)
        if poison == "extreme":
            _scatter_extremes(g, [ds], 0.05)
        elif poison:  # non-finite joint state in a few environments, other ones every step
            o = 16 + 8 * step
            for i, (j, c, v) in enumerate([(0, 0, "nan"), (1, 0, "nan"), (0, 1, "nan"), (1, 1, "nan"), (0, 0, "inf"),
                                           (1, 0, "-inf"), (0, 1, "inf"), (1, 1, "-inf")]):
                ds[o + i, j, c] = float(v)
        t.dof_state.copy_(ds.view(N * 2, 2))
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        st = torch.get_rng_state()
        t.post_physics_step()
        dense = np.zeros((N, 4), np.float32)
        if len(ids) > 0:
            up, uv = _replay_uniforms(st, [(len(ids), 2), (len(ids), 2)])
            dense[ids.numpy(), 0:2] = up.numpy()
            dense[ids.numpy(), 2:4] = uv.numpy()
        rec["rand"].append(dense)
        rec["out_dof_state"].append(t.dof_state.numpy().copy())
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    out["meta"] = _meta(task="Cartpole", N=N, T=T, seed=seed, reset_dist=t.reset_dist,
                        max_episode_length=t.max_episode_length)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_cartpole_sample(seed=5, N=257, T=6):
    """CartpoleSample (tasks/cartpole_sample.py): the fork's trajectory-sampling variant.
    Records its post_physics_step (reward variant :316-335) and the sequence dict that
    sample mode accumulates (:99-107, :289-293)."""
    from collections import defaultdict
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.cartpole_sample")
    cfg = _cfg("CartpoleSample") if os.path.exists(os.path.join(CFG, "CartpoleSample.yaml")) else _cfg("Cartpole")

    class Fake(mod.CartpoleSample):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    t.cfg = cfg
    t.reset_dist = cfg["env"]["resetDist"]
    t.max_episode_length = 500
    t.num_envs, t.num_dof, t.device = N, 2, "cpu"
    t.gym, t.sim = _Any(), None
    t.sample, t.is_train, t.frame_no, t.max_frame_length = True, True, 0, 900
    t.dof_state = torch.zeros(N * 2, 2)
    t.dof_pos = t.dof_state.view(N, 2, 2)[..., 0]
    t.dof_vel = t.dof_state.view(N, 2, 2)[..., 1]
    t.obs_buf = torch.zeros(N, 4)
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.zeros(N, dtype=torch.long)
    t.progress_buf = torch.randint(0, 500, (N,), generator=g)
    t.progress_buf[:4] = torch.tensor([497, 498, 499, 0])

    def draw():
        ds = torch.empty(N, 2, 2)
        ds[:, 0, 0] = (torch.rand(N, generator=g) * 2 - 1) * 3.5
        ds[:, 1, 0] = (torch.rand(N, generator=g) * 2 - 1) * 2.0
        ds[:, :, 1] = torch.randn(N, 2, generator=g)
        return ds

    ds0 = draw()
    t.dof_state.copy_(ds0.view(N * 2, 2))
    t.sequence = defaultdict()
    t.sequence["x"], t.sequence["theta"] = t.get_x_theta(t.compute_observations())
    rec = {k: [] for k in ("in_dof_state", "rand", "out_dof_state", "obs", "rew", "reset", "progress")}
    init = {"reset0": t.reset_buf.numpy().copy(), "progress0": t.progress_buf.numpy().copy(),
            "dof_state0": ds0.numpy().reshape(N * 2, 2).copy()}
    torch.manual_seed(seed + 100)
    for step in range(T):
        t.dof_state.copy_(draw().view(N * 2, 2))
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        st = torch.get_rng_state()
        t.post_physics_step()
        dense = np.zeros((N, 4), np.float32)
        if len(ids) > 0:
            up, uv = _replay_uniforms(st, [(len(ids), 2), (len(ids), 2)])
            dense[ids.numpy(), 0:2] = up.numpy()
            dense[ids.numpy(), 2:4] = uv.numpy()
        rec["rand"].append(dense)
        rec["out_dof_state"].append(t.dof_state.numpy().copy())
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    out["sequence_x"] = t.sequence["x"].numpy().copy()          # [N, T+1]
    out["sequence_theta"] = t.sequence["theta"].numpy().copy()
    out["meta"] = _meta(task="CartpoleSample", N=N, T=T, seed=seed, reset_dist=t.reset_dist,
                        max_episode_length=t.max_episode_length)
    np.savez_compressed(os.path.join(OUT, "cartpole_sample.npz"), **out)
    print("cartpole_sample.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")},
          "resets", int(out["reset"].sum()))


# --------------------------------------------------------------------------
def _loco_common(t, cfg, N, D, g, z_mean):
    e = cfg["env"]
    t.cfg = cfg
    t.max_episode_length = e["episodeLength"]
    t.randomize = False
    t.dof_vel_scale = e["dofVelocityScale"]
    t.contact_force_scale = e["contactForceScale"]
    t.power_scale = e["powerScale"]
    t.heading_weight = e["headingWeight"]
    t.up_weight = e["upWeight"]
    t.actions_cost_scale = e["actionsCost"]
    t.energy_cost_scale = e["energyCost"]
    t.joints_at_limit_cost_scale = e["jointsAtLimitCost"]
    t.death_cost = e["deathCost"]
    t.termination_height = e["terminationHeight"]
    t.debug_viz = False
    t.viewer = None
    t.num_envs, t.num_dof, t.device = N, D, "cpu"
    t.gym, t.sim = _Any(), None
    t.up_axis_idx = 2
    t.extras = {}
    # synthetic joint limits (the asset files are not in the reference tree);
    # one joint with lower > 0 and one with upper < 0 exercise initial_dof_pos
    lo = torch.tensor([-0.5 - 0.05 * j for j in range(D)])
    hi = torch.tensor([0.6 + 0.03 * j for j in range(D)])
    lo[1], hi[1] = 0.1, 0.9
    lo[2], hi[2] = -0.9, -0.1
    t.dof_limits_lower, t.dof_limits_upper = lo, hi
    t.root_states = torch.zeros(N, 13)
    t.initial_root_states = torch.zeros(N, 13)
    t.initial_root_states[:, 0:2] = torch.randn(N, 2, generator=g) * 0.05
    t.initial_root_states[:, 2] = z_mean
    t.initial_root_states[:, 6] = 1.0
    t.dof_state = torch.zeros(N * D, 2)
    t.dof_pos = t.dof_state.view(N, D, 2)[..., 0]
    t.dof_vel = t.dof_state.view(N, D, 2)[..., 1]
    # ant.py:114-117 / humanoid.py:114-117
    zero = torch.tensor([0.0])
    t.initial_dof_pos = torch.zeros(N, D)
    t.initial_dof_pos = torch.where(lo > zero, lo, torch.where(hi < zero, hi, t.initial_dof_pos))
    t.start_rotation = torch.tensor([0.0, 0.0, 0.0, 1.0])
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    t.up_vec = torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1)
    t.heading_vec = torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1)
    t.inv_start_rot = tj.quat_conjugate(t.start_rotation).repeat(N, 1)
    t.basis_vec0 = t.heading_vec.clone()
    t.basis_vec1 = t.up_vec.clone()
    t.targets = torch.tensor([1000.0, 0.0, 0.0]).repeat(N, 1)
    t.dt = cfg["sim"]["dt"]
    t.potentials = torch.tensor([-1000.0 / t.dt]).repeat(N)
    t.prev_potentials = t.potentials.clone()
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.ones(N, dtype=torch.long)
    t.progress_buf = torch.randint(0, 1000, (N,), generator=g)
    t.progress_buf[:6] = torch.tensor([996, 997, 998, 999, 1000, 0])
    t.randomize_buf = torch.zeros(N, dtype=torch.long)


def _loco_inputs(t, N, D, S, g, z_mean, z_std, with_force):
    root = torch.empty(N, 13)
    root[:, 0:3] = torch.randn(N, 3, generator=g) * 0.1
    root[:, 2] = z_mean + torch.randn(N, generator=g) * z_std
    root[:, 3:7] = _rand_quat(g, N)
    root[:, 7:13] = torch.randn(N, 6, generator=g)
    root[3, 2] = t.termination_height  # float32(h) < h(double->f32) is False
    root[4, 0] = 1000.0                # to_target == 0 -> normalize clamps at 1e-9
    root[4, 1] = 0.0
    t.root_states.copy_(root)
    lo, hi = t.dof_limits_lower, t.dof_limits_upper
    u = torch.rand(N, D, generator=g) * 1.06 - 0.03  # a few % beyond the limits
    pos = lo + (hi - lo) * u
    vel = torch.randn(N, D, generator=g)
    t.dof_state.view(N, D, 2)[..., 0] = pos
    t.dof_state.view(N, D, 2)[..., 1] = vel
    t.vec_sensor_tensor = torch.randn(N, S, generator=g) * 10
    t.actions = torch.rand(N, D, generator=g) * 2 - 1
    if with_force:
        t.dof_force_tensor = torch.randn(N, D, generator=g) * 30


def _poison_loco(t, step, humanoid):
    """Non-finite simulator state (a physics blow-up) in a few environments, other ones every step: what the
    reference's own arithmetic makes of NaN / +-Inf (torch's comparisons, where, clamp, atan2) is the contract."""
    nan, inf = float("nan"), float("inf")
    o = 10 + 16 * step
    r, ds = t.root_states, t.dof_state.view(t.num_envs, t.num_dof, 2)
    r[o + 0, 0] = nan            # position x
    r[o + 1, 6] = nan            # quaternion w
    r[o + 2, 2] = inf            # height
    r[o + 3, 2] = -inf           # height below everything: termination
    r[o + 4, 7] = inf            # linear velocity
    r[o + 5, 11] = -inf          # angular velocity
    ds[o + 6, 3, 0] = nan        # a joint position
    ds[o + 7, 2, 1] = -inf       # a joint velocity
    t.actions[o + 8, 1] = nan
    t.actions[o + 9, 0] = inf
    t.vec_sensor_tensor[o + 10, 5] = nan
    r[o + 11, 3:7] = 0.0         # a zero quaternion
    r[o + 12, 3:7] = torch.tensor([nan, nan, nan, nan])
    r[o + 13, 0] = inf           # infinitely far from the target
    if humanoid:
        t.dof_force_tensor[o + 14, 4] = nan
        t.dof_force_tensor[o + 15, 0] = inf


def _scatter_extremes(g, tensors, frac=0.02):
    """Huge, tiny, subnormal, signed-zero and non-finite values over about `frac` of the entries of every given
    tensor (in place): what the reference's arithmetic makes of them is the contract, as for NaN / Inf."""
    specials = torch.tensor([float("nan"), float("inf"), -float("inf"), 3e38, -3e38, 1e30, -1e30, 1e-30, 1e-40, -1e-45,
                             0.0, -0.0, 1e19, -1e19, 1e-20], dtype=torch.float)
    for x in tensors:
        flat = x.view(-1)
        m = max(1, int(flat.numel() * frac))
        idx = torch.randint(0, flat.numel(), (m,), generator=g)
        flat[idx] = specials[torch.randint(0, len(specials), (m,), generator=g)]


def _scale_quats(g, q):
    """Quaternions that are not of unit length (in place): a simulator's drift, or a caller that never normalised;
    the factor of each is drawn from 0.5 ... 2."""
    q *= torch.rand(q.shape[0], 1, generator=g) * 1.5 + 0.5


def _gen_loco(name, modname, clsname, cfgname, D, S, humanoid, seed, N, T, env_over=None, dt=None, targets=None,
              start_rotation=None, poison=False):
    """env_over / dt / targets / start_rotation: non-default values, so that a recording also pins how every
    parameter enters the arithmetic (the default files only exercise the YAML defaults)."""
    _install_gym_stubs()
    mod = ref_shim.load(modname)
    cfg = _cfg(cfgname)
    cfg["env"].update(env_over or {})
    if dt is not None:
        cfg["sim"]["dt"] = dt
    Base = getattr(mod, clsname)

    class Fake(Base):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    z_mean, z_std = (1.34, 0.3) if humanoid else (0.55, 0.1)
    _loco_common(t, cfg, N, D, g, z_mean)
    if targets is not None:
        t.targets = torch.tensor(targets, dtype=torch.float).repeat(N, 1)
        t.potentials = -torch.norm((t.targets - t.initial_root_states[:, 0:3]) * torch.tensor([1.0, 1.0, 0.0]), p=2, dim=-1) / t.dt
        t.prev_potentials = t.potentials.clone()
    if start_rotation is not None:
        tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
        q = torch.tensor(start_rotation, dtype=torch.float)
        t.start_rotation = q / q.norm()
        t.inv_start_rot = tj.quat_conjugate(t.start_rotation).repeat(N, 1)
    nobs = 108 if humanoid else 60
    t.obs_buf = torch.zeros(N, nobs)
    if humanoid:
        t.angular_velocity_scale = cfg["env"].get("angularVelocityScale", 0.1)
        efforts = (torch.rand(D, generator=g) * 180 + 20).tolist()
        t.max_motor_effort = max(efforts)
        t.motor_efforts = torch.tensor(efforts)

    keys = ["in_root", "in_dof_state", "in_sensors", "in_actions", "rand", "out_dof_state", "obs",
            "potentials", "prev_potentials", "up_vec", "heading_vec", "rew", "reset", "progress",
            "randomize"]
    if humanoid:
        keys.append("in_dof_force")
    else:
        keys.append("true_objective")
    rec = {k: [] for k in keys}
    init = {"reset0": t.reset_buf.numpy().copy(), "progress0": t.progress_buf.numpy().copy(),
            "potentials0": t.potentials.numpy().copy(),
            "initial_root_states": t.initial_root_states.numpy().copy(),
            "initial_dof_pos": t.initial_dof_pos.numpy().copy(),
            "dof_limits_lower": t.dof_limits_lower.numpy().copy(),
            "dof_limits_upper": t.dof_limits_upper.numpy().copy(),
            "targets": t.targets[0].numpy().copy(), "inv_start_rot": t.inv_start_rot[0].numpy().copy()}
    if humanoid:
        init["motor_efforts"] = t.motor_efforts.numpy().copy()
    torch.manual_seed(seed + 100)
    for step in range(T):
        _loco_inputs(t, N, D, S, g, z_mean, z_std, humanoid)
        if poison == "extreme":
            _scatter_extremes(g, [t.root_states, t.dof_state, t.vec_sensor_tensor, t.actions] +
                              ([t.dof_force_tensor] if humanoid else []))
        elif poison == "nonunit":
            _scale_quats(g, t.root_states[:, 3:7])
        elif poison:
            _poison_loco(t, step, humanoid)
        rec["in_root"].append(t.root_states.numpy().copy())
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        rec["in_sensors"].append(t.vec_sensor_tensor.numpy().copy())
        rec["in_actions"].append(t.actions.numpy().copy())
        if humanoid:
            rec["in_dof_force"].append(t.dof_force_tensor.numpy().copy())
        ids = t.reset_buf.nonzero(as_tuple=False).flatten()
        st = torch.get_rng_state()
        t.post_physics_step()
        dense = np.zeros((N, 2 * D), np.float32)
        if len(ids) > 0:
            up, uv = _replay_uniforms(st, [(len(ids), D), (len(ids), D)])
            dense[ids.numpy(), :D] = up.numpy()
            dense[ids.numpy(), D:] = uv.numpy()
        rec["rand"].append(dense)
        rec["out_dof_state"].append(t.dof_state.numpy().copy())
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["potentials"].append(t.potentials.numpy().copy())
        rec["prev_potentials"].append(t.prev_potentials.numpy().copy())
        rec["up_vec"].append(t.up_vec.numpy().copy())
        rec["heading_vec"].append(t.heading_vec.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
        rec["randomize"].append(t.randomize_buf.numpy().copy())
        if not humanoid:
            rec["true_objective"].append(t.extras["true_objective"].numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    e = cfg["env"]
    meta = dict(task=cfgname, N=N, T=T, seed=seed, num_dof=D, num_sensor_floats=S, dt=cfg["sim"]["dt"],
                dof_vel_scale=e["dofVelocityScale"], contact_force_scale=e["contactForceScale"],
                angular_velocity_scale=e.get("angularVelocityScale", 0.1),
                up_weight=e["upWeight"], heading_weight=e["headingWeight"],
                actions_cost_scale=e["actionsCost"], energy_cost_scale=e["energyCost"],
                joints_at_limit_cost_scale=e["jointsAtLimitCost"],
                termination_height=e["terminationHeight"], death_cost=e["deathCost"],
                max_episode_length=e["episodeLength"])
    if humanoid:
        meta["max_motor_effort"] = t.max_motor_effort
    out["meta"] = _meta(**meta)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_ant_imitate(seed=41, N=257, T=7, F=9, name="ant_imitate", poison=False):
    """AntImitate (tasks/ant_imitate.py): Ant observations + the pose-imitation reward with its
    global quantile reset threshold (:447-555) and the target_pose_idx bookkeeping (:405-419)."""
    import contextlib
    import io
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.ant_imitate")
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    cfg = _cfg("AntImitate")

    class Fake(mod.AntImitate):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    D, S = 8, 24
    _loco_common(t, cfg, N, D, g, 0.55)
    t.obs_buf = torch.zeros(N, 60)
    lo, hi = t.dof_limits_lower, t.dof_limits_upper
    # np.loadtxt gives float64: the reward arithmetic of the reference runs in double
    motion = (lo.double() + (hi - lo).double() * torch.rand(F, D, generator=g, dtype=torch.float64)).numpy()
    t.motion_to_imitate = torch.tensor(motion)
    t.target_pose_idx = torch.ones(N, dtype=torch.long)
    t.reset_buf = (torch.rand(N, generator=g) < 0.3).long()
    keys = ["in_root", "in_dof_state", "in_sensors", "in_actions", "rand", "out_dof_state", "obs", "potentials",
            "prev_potentials", "rew", "reset", "progress", "randomize", "target_pose_idx", "pose_diff", "threshold",
            "in_reset"]
    rec = {k: [] for k in keys}
    init = {"reset0": t.reset_buf.numpy().copy(), "progress0": t.progress_buf.numpy().copy(),
            "potentials0": t.potentials.numpy().copy(), "target_pose_idx0": t.target_pose_idx.numpy().copy(),
            "initial_root_states": t.initial_root_states.numpy().copy(),
            "initial_dof_pos": t.initial_dof_pos.numpy().copy(),
            "dof_limits_lower": lo.numpy().copy(), "dof_limits_upper": hi.numpy().copy(), "motion": motion,
            "targets": t.targets[0].numpy().copy(), "inv_start_rot": t.inv_start_rot[0].numpy().copy()}
    torch.manual_seed(seed + 100)
    for step in range(T):
        _loco_inputs(t, N, D, S, g, 0.55, 0.1, False)
        if step in (2, 5):  # joints close to the target frame: the 0.99 quantile drops below the 1.8 floor
            tgt = t.motion_to_imitate[(t.target_pose_idx + 1).clamp(max=F - 1)].float()
            t.dof_state.view(N, D, 2)[..., 0] = tgt + 0.02 * torch.randn(N, D, generator=g)
            t.reset_buf.zero_()  # no re-drawn joints in these steps
        if poison == "extreme":
            _scatter_extremes(g, [t.root_states, t.dof_state, t.vec_sensor_tensor, t.actions], 0.01)
        rec["in_reset"].append(t.reset_buf.numpy().copy())
        rec["in_root"].append(t.root_states.numpy().copy())
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        rec["in_sensors"].append(t.vec_sensor_tensor.numpy().copy())
        rec["in_actions"].append(t.actions.numpy().copy())
        ids = t.reset_buf.nonzero(as_tuple=False).flatten()
        st = torch.get_rng_state()
        idx_used = (t.target_pose_idx + 1).clone()
        with contextlib.redirect_stdout(io.StringIO()):  # the reference prints from inside the jit function
            t.post_physics_step()
        dense = np.zeros((N, 2 * D), np.float32)
        if len(ids) > 0:
            up, uv = _replay_uniforms(st, [(len(ids), D), (len(ids), D)])
            dense[ids.numpy(), :D] = up.numpy()
            dense[ids.numpy(), D:] = uv.numpy()
        rec["rand"].append(dense)
        # the intermediate the reward function does not return, by its own expressions (:508-510, :529)
        pose_scaled = tj.unscale(t.motion_to_imitate[idx_used], lo, hi)
        pose_diff = torch.sum(torch.abs(t.obs_buf[:, 12:20] - pose_scaled), dim=-1)
        thr = max(1.8, torch.quantile(pose_diff, 0.99))
        rec["pose_diff"].append(pose_diff.numpy().copy())
        rec["threshold"].append(np.float64(float(thr)))
        rec["out_dof_state"].append(t.dof_state.numpy().copy())
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["potentials"].append(t.potentials.numpy().copy())
        rec["prev_potentials"].append(t.prev_potentials.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
        rec["randomize"].append(t.randomize_buf.numpy().copy())
        rec["target_pose_idx"].append(t.target_pose_idx.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    e = cfg["env"]
    out["meta"] = _meta(task="AntImitate", N=N, T=T, F=F, seed=seed, num_dof=D, num_sensor_floats=S, dt=cfg["sim"]["dt"],
                        dof_vel_scale=e["dofVelocityScale"], contact_force_scale=e["contactForceScale"],
                        angular_velocity_scale=0.1, up_weight=e["upWeight"], heading_weight=e["headingWeight"],
                        actions_cost_scale=e["actionsCost"], energy_cost_scale=e["energyCost"],
                        joints_at_limit_cost_scale=e["jointsAtLimitCost"], termination_height=e["terminationHeight"],
                        death_cost=e["deathCost"], max_episode_length=e["episodeLength"], quantile=0.99,
                        distance_floor=1.8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")},
          "thresholds", out["threshold"], "resets", out["reset"].sum(1), "idx max", out["target_pose_idx"].max(1),
          "success", [(out["rew"][i] == 6.0).sum() for i in range(T)])


def gen_humanoid_imitate(seed=51, N=67, T=4, F=48, name="humanoid_imitate", poison=False):
    """HumanoidImitate (tasks/humanoid_imitate.py): 63-dof humanoid, 205-wide observation (:795-831), pose /
    rigid-body tracking reward against frame 41 of the motion file (:672-758).
    Run with PYTORCH_JIT=0: extract_positions_from_rigid_body_states (utils/torch_jit_utils.py:670-681) builds its
    index tensor on "cuda:0"; un-scripted, that one allocation can be redirected to the CPU while every other line
    of the reference runs as written."""
    import contextlib
    import io
    assert os.environ.get("PYTORCH_JIT") == "0", "run as: PYTORCH_JIT=0 python oracle/gen_golden.py humanoid_imitate"
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.humanoid_imitate")
    cfg = _cfg("HumanoidImitate")

    class Fake(mod.HumanoidImitate):
        def __init__(self):
            pass

    real_empty = torch.empty

    def cpu_empty(*a, **k):
        if str(k.get("device", "")) == "cuda:0":
            k["device"] = "cpu"
        return real_empty(*a, **k)

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    D, S, NBODY = 63, 12, 22
    _loco_common(t, cfg, N, D, g, 0.9)
    e = cfg["env"]
    t.termination_radius = e["terminationRadius"]
    t.angular_velocity_scale = e.get("angularVelocityScale", 0.1)
    t.obs_buf = torch.zeros(N, 205)
    efforts = (torch.rand(D, generator=g) * 180 + 20).tolist()
    t.max_motor_effort = max(efforts)
    t.motor_efforts = torch.tensor(efforts)
    t.control_steps, t.debug_viz_rigid_bodies = 0, False
    lo, hi = t.dof_limits_lower, t.dof_limits_upper
    motion = np.concatenate([(lo.double() + (hi - lo).double() * torch.rand(F, D, generator=g, dtype=torch.float64)).numpy(),
                             torch.randn(F, NBODY * 13, generator=g, dtype=torch.float64).numpy() * 0.5], axis=1)
    t.motion_to_imitate = torch.tensor(motion)
    t.rigid_body_states = torch.zeros(N, NBODY * 13)
    t.reset_buf = (torch.rand(N, generator=g) < 0.3).long()
    t.progress_buf[:4] = torch.tensor([997, 998, 999, 5])
    keys = ["in_root", "in_dof_state", "in_rigid_body_states", "in_actions", "rand", "out_dof_state", "obs",
            "potentials", "prev_potentials", "rew", "reset", "progress", "randomize"]
    rec = {k: [] for k in keys}
    init = {"reset0": t.reset_buf.numpy().copy(), "progress0": t.progress_buf.numpy().copy(),
            "potentials0": t.potentials.numpy().copy(), "initial_root_states": t.initial_root_states.numpy().copy(),
            "initial_dof_pos": t.initial_dof_pos.numpy().copy(), "dof_limits_lower": lo.numpy().copy(),
            "dof_limits_upper": hi.numpy().copy(), "motor_efforts": t.motor_efforts.numpy().copy(), "motion": motion}
    torch.manual_seed(seed + 100)
    for step in range(T):
        _loco_inputs(t, N, D, S, g, 0.9, 0.4, False)
        t.root_states[:, 0:2] = torch.randn(N, 2, generator=g) * 0.6   # some leave the termination radius
        t.root_states[5, 0:2] = torch.tensor([1.0, 0.0])               # exactly on the radius: not > r^2
        if step == 2:  # joints near the target frame: large pose reward
            t.dof_state.view(N, D, 2)[..., 0] = t.motion_to_imitate[41, :D].float() + 0.01 * torch.randn(N, D, generator=g)
        t.rigid_body_states.copy_(torch.randn(N, NBODY * 13, generator=g) * 0.5)
        if poison == "extreme":
            _scatter_extremes(g, [t.root_states, t.dof_state, t.rigid_body_states, t.actions], 0.004)
        rec["in_root"].append(t.root_states.numpy().copy())
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        rec["in_rigid_body_states"].append(t.rigid_body_states.numpy().copy())
        rec["in_actions"].append(t.actions.numpy().copy())
        ids = t.reset_buf.nonzero(as_tuple=False).flatten()
        st = torch.get_rng_state()
        torch.empty = cpu_empty
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                t.post_physics_step()
        finally:
            torch.empty = real_empty
        dense = np.zeros((N, 2 * D), np.float32)
        if len(ids) > 0:
            up, uv = _replay_uniforms(st, [(len(ids), D), (len(ids), D)])
            dense[ids.numpy(), :D] = up.numpy()
            dense[ids.numpy(), D:] = uv.numpy()
        rec["rand"].append(dense)
        rec["out_dof_state"].append(t.dof_state.numpy().copy())
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["potentials"].append(t.potentials.numpy().copy())
        rec["prev_potentials"].append(t.prev_potentials.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
        rec["randomize"].append(t.randomize_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    out["meta"] = _meta(task="HumanoidImitate", N=N, T=T, F=F, seed=seed, num_dof=D, num_bodies=NBODY, dt=cfg["sim"]["dt"],
                        dof_vel_scale=e["dofVelocityScale"], up_weight=e["upWeight"], actions_cost_scale=e["actionsCost"],
                        termination_height=e["terminationHeight"], termination_radius=e["terminationRadius"],
                        max_episode_length=e["episodeLength"], max_motor_effort=t.max_motor_effort, pose_frame=41)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")},
          "resets", out["reset"].sum(1), "rew", out["rew"].mean(1))


def gen_ant(seed=1, N=449, T=4):
    _gen_loco("ant", "isaacgymenvs.tasks.ant", "Ant", "Ant", 8, 24, False, seed, N, T)


def gen_humanoid(seed=2, N=211, T=4):
    _gen_loco("humanoid", "isaacgymenvs.tasks.humanoid", "Humanoid", "Humanoid", 21, 12, True, seed, N, T)


ALT_ENV = dict(dofVelocityScale=0.37, contactForceScale=0.023, upWeight=0.17, headingWeight=0.61, actionsCost=0.013,
               energyCost=0.071, jointsAtLimitCost=0.29, deathCost=-3.5, episodeLength=700)


def gen_loco_alt():
    """Ant and Humanoid with every tunable off its YAML default, another time step, another target and a
    tilted start rotation."""
    _gen_loco("ant_alt", "isaacgymenvs.tasks.ant", "Ant", "Ant", 8, 24, False, 71, 150, 3,
              env_over=dict(ALT_ENV, terminationHeight=0.47), dt=0.0125, targets=[-300.0, 450.0, 2.0],
              start_rotation=[0.1, -0.2, 0.3, 0.9])
    _gen_loco("humanoid_alt", "isaacgymenvs.tasks.humanoid", "Humanoid", "Humanoid", 21, 12, True, 72, 90, 3,
              env_over=dict(ALT_ENV, terminationHeight=1.1, angularVelocityScale=0.4), dt=0.02,
              targets=[120.0, -80.0, 0.0], start_rotation=[0.0, 0.3, 0.0, 0.95])


def gen_loco_nonfinite():
    """Ant and Humanoid with NaN / +-Inf in the simulator tensors of some environments."""
    _gen_loco("ant_nonfinite", "isaacgymenvs.tasks.ant", "Ant", "Ant", 8, 24, False, 91, 96, 4, poison=True)
    _gen_loco("humanoid_nonfinite", "isaacgymenvs.tasks.humanoid", "Humanoid", "Humanoid", 21, 12, True, 92, 96, 4, poison=True)


def gen_loco_extreme():
    """Ant and Humanoid with huge / tiny / subnormal / signed-zero / non-finite values scattered over 2 % of the
    simulator state and the actions."""
    _gen_loco("ant_extreme", "isaacgymenvs.tasks.ant", "Ant", "Ant", 8, 24, False, 93, 192, 3, poison="extreme")
    _gen_loco("humanoid_extreme", "isaacgymenvs.tasks.humanoid", "Humanoid", "Humanoid", 21, 12, True, 94, 192, 3, poison="extreme")


# --------------------------------------------------------------------------
def _poison_rows(x):
    """NaN / +Inf / -Inf, one entry per row, in rows 100 ... of a [n] or [n, k] input (in place; returns x)."""
    vals = [float("nan"), float("inf"), -float("inf")]
    if x.dim() == 1:
        for i in range(6):
            x[100 + i] = vals[i % 3]
    else:
        k = x.shape[1]
        for i in range(3 * k):
            x[100 + i, i % k] = vals[(i // k) % 3]
        x[100 + 3 * k] = float("nan")
    return x


def gen_math(seed=3, n=1536, name="math", P=lambda x: x):
    """torch_jit_utils functions on random inputs plus the edge cases of each."""
    _install_gym_stubs()
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    g = torch.Generator().manual_seed(seed)

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True)

    qa = unit(torch.randn(n, 4, generator=g))
    qb = unit(torch.randn(n, 4, generator=g))
    qa[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    qb[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    v = torch.randn(n, 3, generator=g)
    v[1] = 0.0
    P(qa), P(qb), P(v)
    out = {"qa": qa, "qb": qb, "v": v}
    out["quat_mul"] = tj.quat_mul(qa, qb)
    out["quat_conjugate"] = tj.quat_conjugate(qa)
    out["quat_rotate"] = tj.quat_rotate(qa, v)
    out["quat_rotate_inverse"] = tj.quat_rotate_inverse(qa, v)

    # slerp: generic, antipodal (dot < 0), identical (|cos| >= 1), nearly
    # parallel (sin < 0.001), blend outside [0, 1]
    s0 = qa.clone()
    s1 = qb.clone()
    s1[10:20] = -s0[10:20]
    s1[20:30] = s0[20:30]
    s1[30:60] = unit(s0[30:60] + 1e-4 * torch.randn(30, 4, generator=g))
    t = torch.rand(n, 1, generator=g)
    t[60:90] = torch.rand(30, 1, generator=g) * 3 - 1.5
    t[0] = 0.0
    t[1] = 1.0
    P(t)
    out.update(slerp_q0=s0, slerp_q1=s1, slerp_t=t, slerp=tj.slerp(s0, s1, t))

    angle = (torch.rand(n, generator=g) * 2 - 1) * 4.0
    axis = torch.randn(n, 3, generator=g)
    angle[0] = 0.0
    axis[1] = 0.0  # zero-norm axis: clamp(min=1e-9) path
    P(angle), P(axis)
    out.update(angle=angle, axis=axis, quat_from_angle_axis=tj.quat_from_angle_axis(angle, axis))

    x = (torch.rand(n, generator=g) * 2 - 1) * 10.0
    x[:5] = torch.tensor([0.0, math.pi, -math.pi, 2 * math.pi, -0.0])
    P(x)
    out.update(na_x=x, normalize_angle=tj.normalize_angle(x))

    qe = qa.clone()
    qe[2] = torch.tensor([0.0, 0.0, 0.0, 1.0])   # sin_theta == 0 -> default axis
    qe[3] = torch.tensor([0.0, 0.0, 0.0, -1.0])
    qe[4] = unit(torch.tensor([1e-6, 0.0, 0.0, 1.0]))
    out.update(qe=qe, quat_to_exp_map=tj.quat_to_exp_map(qe))

    e = torch.randn(n, 3, generator=g) * 1.5
    e[0] = 0.0                                   # 0 / 0 lanes, masked by where
    e[1] = torch.tensor([1e-7, 0.0, 0.0])
    e[2] = torch.tensor([4.0, 0.0, 0.0])         # angle > pi wraps
    P(e)
    out.update(e=e, exp_map_to_quat=tj.exp_map_to_quat(e))

    out["quat_to_tan_norm"] = tj.quat_to_tan_norm(qa)
    out["calc_heading_quat_inv"] = tj.calc_heading_quat_inv(qa)

    d = 16
    lo = -torch.rand(d, generator=g) - 0.2
    hi = torch.rand(d, generator=g) + 0.3
    xs = P(torch.randn(n, d, generator=g))
    out.update(aff_x=xs, aff_lo=lo, aff_hi=hi, scale=tj.scale(xs, lo, hi), unscale=tj.unscale(xs, lo, hi),
               tensor_clamp=tj.tensor_clamp(xs, lo, hi))
    out = {k: v.numpy() for k, v in out.items()}
    out["meta"] = _meta(task="torch_jit_utils", n=n, seed=seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_math2(seed=4, n=1024, name="math2", P=lambda x: x):
    """The rest of torch_jit_utils: every function of the module that works on [n, k] rows."""
    _install_gym_stubs()
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    g = torch.Generator().manual_seed(seed)

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True)

    qa = unit(torch.randn(n, 4, generator=g))
    qb = unit(torch.randn(n, 4, generator=g))
    qa[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])
    qa[1] = torch.tensor([0.0, 0.70710678, 0.0, 0.70710678])    # pitch = +90 degrees: |sinp| >= 1 branch
    qa[2] = torch.tensor([0.0, -0.70710678, 0.0, 0.70710678])
    qa[3] = torch.tensor([0.0, 0.0, 0.0, -1.0])
    v = torch.randn(n, 3, generator=g)
    v2 = torch.randn(n, 3, generator=g)
    v[4] = 0.0
    P(qa), P(qb), P(v), P(v2)
    out = {"qa": qa, "qb": qb, "v": v, "v2": v2}

    raw4 = torch.randn(n, 4, generator=g) * 3.0
    raw4[0] = 0.0                                               # clamp(min=1e-9) path
    P(raw4)
    out.update(raw4=raw4, normalize3=tj.normalize(v), normalize4=tj.normalize(raw4), quat_unit=tj.quat_unit(raw4))
    out["quat_apply"] = tj.quat_apply(qa, v)
    qi, ti = tj.tf_inverse(qa, v)
    out.update(tf_inverse_q=qi, tf_inverse_t=ti, tf_apply=tj.tf_apply(qa, v, v2), tf_vector=tj.tf_vector(qa, v))
    qc, tc = tj.tf_combine(qa, v, qb, v2)
    out.update(tf_combine_q=qc, tf_combine_t=tc, get_basis_vector=tj.get_basis_vector(qa, v))

    cs_b = torch.randn(n, generator=g)
    cs_b[:4] = torch.tensor([0.0, -0.0, float("nan"), float("inf")])
    out.update(cs_b=cs_b, copysign_pos=tj.copysign(math.pi / 2.0, cs_b), copysign_neg=tj.copysign(-2.5, cs_b))

    r, pch, y = tj.get_euler_xyz(qa)
    out.update(euler_roll=r, euler_pitch=pch, euler_yaw=y)
    er = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    ep = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    ey = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    er[0] = ep[0] = ey[0] = 0.0
    P(er), P(ep), P(ey)
    out.update(er=er, ep=ep, ey=ey, quat_from_euler_xyz=tj.quat_from_euler_xyz(er, ep, ey),
               euler_xyz_to_exp_map=tj.euler_xyz_to_exp_map(er, ep, ey))

    torch.manual_seed(1234)
    u = torch.rand(n, 6)
    torch.manual_seed(1234)
    out.update(rand_u=u, torch_rand_float=tj.torch_rand_float(-0.2, 0.35, (n, 6), "cpu"))
    assert torch.equal(out["torch_rand_float"], (0.35 - -0.2) * u + -0.2)

    isr = unit(torch.randn(n, 4, generator=g))
    to_target = torch.randn(n, 3, generator=g) * 10.0
    to_target[:, 2] = 0.0
    to_target[5] = 0.0
    vec0 = torch.randn(n, 3, generator=g)
    vec1 = torch.randn(n, 3, generator=g)
    P(isr), P(to_target), P(vec0), P(vec1)
    for idx in (2, 1):
        tq, upp, hp, upv, hv = tj.compute_heading_and_up(qa, isr, to_target, vec0, vec1, idx)
        out.update({"hu%d_torso_quat" % idx: tq, "hu%d_up_proj" % idx: upp, "hu%d_heading_proj" % idx: hp,
                    "hu%d_up_vec" % idx: upv, "hu%d_heading_vec" % idx: hv})
    out.update(isr=isr, to_target=to_target, vec0=vec0, vec1=vec1)
    targets = P(torch.randn(n, 3, generator=g) * 100.0)
    vl, al, rr, pp, yy, att = tj.compute_rot(qb, v, v2, targets, to_target)
    out.update(targets=targets, rot_vel_loc=vl, rot_angvel_loc=al, rot_roll=rr, rot_pitch=pp, rot_yaw=yy,
               rot_angle_to_target=att)
    for ax in range(3):
        out["quat_axis%d" % ax] = tj.quat_axis(qa, ax)

    d = 12
    lo = -torch.rand(d, generator=g) - 0.2
    hi = torch.rand(d, generator=g) + 0.3
    xs = P(torch.randn(n, d, generator=g))
    out.update(aff_x=xs, aff_lo=lo, aff_hi=hi, scale_transform=tj.scale_transform(xs, lo, hi),
               unscale_transform=tj.unscale_transform(xs, lo, hi), saturate=tj.saturate(xs, lo, hi))

    da = qa.clone()
    db = qb.clone()
    db[10:20] = da[10:20]            # identical: distance 0
    db[20:30] = -da[20:30]           # antipodal: same rotation
    da[30:40] = da[30:40] * 1.5      # not unit: the clamp(max=1) branch
    out.update(diff_a=da, diff_b=db, quat_diff_rad=tj.quat_diff_rad(da, db))

    pose = P(torch.cat([torch.randn(n, 3, generator=g), qb], dim=-1))
    out.update(pose=pose, local_to_world_space=tj.local_to_world_space(v, pose))

    qe = qa.clone()
    qe[4] = unit(torch.tensor([1e-6, 0.0, 0.0, 1.0]))
    ang, axs = tj.quat_to_angle_axis(qe)
    out.update(qe=qe, q2aa_angle=ang, q2aa_axis=axs, angle_axis_to_exp_map=tj.angle_axis_to_exp_map(ang, axs))
    e = torch.randn(n, 3, generator=g) * 1.5
    e[0] = 0.0
    e[1] = torch.tensor([1e-7, 0.0, 0.0])
    e[2] = torch.tensor([4.0, 0.0, 0.0])
    P(e)
    ang2, axs2 = tj.exp_map_to_angle_axis(e)
    out.update(e=e, e2aa_angle=ang2, e2aa_axis=axs2)
    out.update(calc_heading=tj.calc_heading(qa), calc_heading_quat=tj.calc_heading_quat(qa))

    out = {k: v.numpy() for k, v in out.items()}
    out["meta"] = _meta(task="torch_jit_utils (remaining functions)", n=n, seed=seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


# --------------------------------------------------------------------------
def success_threshold_c(tol):
    """Largest float32 c in [0, 1] with |2*asin(c)| <= tol under torch's CPU asin
    (bisection over the float32 bit patterns; asin is monotone)."""
    tol32 = torch.tensor(float(tol), dtype=torch.float32)

    def ok(bits):
        c = torch.tensor(np.array([bits], np.uint32).view(np.float32))
        return bool((torch.abs(2.0 * torch.asin(torch.clamp(c, max=1.0))) <= tol32).item())

    lo, hi = 0, int(np.array([1.0], np.float32).view(np.uint32)[0])
    if ok(hi):
        return 1.0
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if ok(mid):
            lo = mid
        else:
            hi = mid
    return float(np.array([lo], np.uint32).view(np.float32)[0])


def _gen_allegro(name, N, T, seed, obs_type="full_state", asymmetric=False, object_type="block",
                 max_consecutive_successes=0, use_relative_control=False, act_moving_average=None,
                 force_scale=None, force_prob_range=(0.001, 0.1), poison=False):
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.allegro_hand")
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    cfg = _cfg("AllegroHand")
    e = cfg["env"]

    class Fake(mod.AllegroHand):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    D = 16
    t.cfg = cfg
    t.device = "cpu"
    t.num_envs = N
    t.gym, t.sim, t.viewer, t.debug_viz = _Any(), None, None, False
    t.dist_reward_scale = e["distRewardScale"]
    t.rot_reward_scale = e["rotRewardScale"]
    t.action_penalty_scale = e["actionPenaltyScale"]
    t.success_tolerance = e["successTolerance"]
    t.reach_goal_bonus = e["reachGoalBonus"]
    t.fall_dist = e["fallDistance"]
    t.fall_penalty = e["fallPenalty"]
    t.rot_eps = e["rotEps"]
    t.vel_obs_scale = 0.2
    t.force_torque_obs_scale = 10.0
    t.reset_position_noise = e["resetPositionNoise"]
    t.reset_rotation_noise = e["resetRotationNoise"]
    t.reset_dof_pos_noise = e["resetDofPosRandomInterval"]
    t.reset_dof_vel_noise = e["resetDofVelRandomInterval"]
    t.force_scale = e.get("forceScale", 0.0) if force_scale is None else force_scale
    t.shadow_hand_dof_speed_scale = e["dofSpeedScale"]
    t.use_relative_control = use_relative_control
    t.act_moving_average = e["actionsMovingAverage"] if act_moving_average is None else act_moving_average
    t.max_episode_length = e["episodeLength"]
    t.print_success_stat = False
    t.max_consecutive_successes = max_consecutive_successes
    t.av_factor = torch.tensor(e.get("averFactor", 0.1), dtype=torch.float)
    t.object_type = object_type
    t.obs_type = obs_type
    t.asymmetric_obs = asymmetric
    t.up_axis_idx = 2
    t.dt = cfg["sim"]["dt"] if "sim" in cfg and "dt" in cfg["sim"] else 0.01667
    t.num_shadow_hand_dofs = D
    t.num_actions = D
    t.extras = {}
    nobs = {"full_no_vel": 50, "full": 72, "full_state": 88}[obs_type]
    lo = -(torch.rand(D, generator=g) * 0.5 + 0.2)
    hi = torch.rand(D, generator=g) * 1.2 + 0.3
    t.shadow_hand_dof_lower_limits, t.shadow_hand_dof_upper_limits = lo, hi
    t.shadow_hand_dof_default_pos = torch.zeros(D)
    t.shadow_hand_dof_default_vel = torch.zeros(D)
    t.shadow_hand_default_dof_pos = torch.zeros(D)
    t.actuated_dof_indices = torch.arange(D)
    t.dof_state = torch.zeros(N * D, 2)
    t.shadow_hand_dof_state = t.dof_state.view(N, -1, 2)[:, :D]
    t.shadow_hand_dof_pos = t.shadow_hand_dof_state[..., 0]
    t.shadow_hand_dof_vel = t.shadow_hand_dof_state[..., 1]
    t.dof_force_tensor = torch.zeros(N, D)
    t.root_state_tensor = torch.zeros(3 * N, 13)
    t.hand_indices = torch.arange(N) * 3
    t.object_indices = torch.arange(N) * 3 + 1
    t.goal_object_indices = torch.arange(N) * 3 + 2
    t.object_init_state = torch.zeros(N, 13)
    t.object_init_state[:, 0:3] = torch.tensor([0.0, -0.2, 0.56]) + 0.01 * torch.randn(N, 3, generator=g)
    t.object_init_state[:, 6] = 1.0
    t.goal_states = t.object_init_state.clone()
    t.goal_states[:, 2] -= 0.04
    t.goal_init_state = t.goal_states.clone()
    t.goal_displacement_tensor = torch.tensor([-0.2, -0.06, 0.12])
    t.prev_targets = torch.zeros(N, D)
    t.cur_targets = torch.zeros(N, D)
    t.x_unit_tensor = torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1)
    t.y_unit_tensor = torch.tensor([0.0, 1.0, 0.0]).repeat(N, 1)
    t.z_unit_tensor = torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1)
    t.obs_buf = torch.zeros(N, nobs)
    t.states_buf = torch.zeros(N, 88 if asymmetric else 0)
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.ones(N, dtype=torch.long)
    t.reset_goal_buf = t.reset_buf.clone()
    t.progress_buf = torch.zeros(N, dtype=torch.long)
    t.randomize_buf = torch.zeros(N, dtype=torch.long)
    t.successes = torch.zeros(N)
    t.consecutive_successes = torch.zeros(1)
    t.force_decay = torch.tensor(0.99)
    t.force_prob_range = torch.tensor([float(force_prob_range[0]), float(force_prob_range[1])])
    t.force_decay_interval = e.get("forceDecayInterval", 0.08)
    t.random_force_prob = torch.zeros(N)
    t.rb_forces = torch.zeros(N, 4, 3)
    t.object_rb_handles = torch.tensor([2])
    t.object_rb_masses = torch.tensor([0.1])

    tol = t.success_tolerance * (2.0 if object_type == "pen" else 1.0)
    cstar = success_threshold_c(tol)
    keys = ["actions", "rand", "pre_dof_state", "pre_root", "pre_goal_states", "pre_prev_targets", "pre_cur_targets",
            "pre_reset", "pre_reset_goal", "pre_progress", "pre_successes",
            "in_dof_state", "in_dof_force", "in_root", "obs", "states", "rew", "reset", "reset_goal", "progress",
            "successes", "consecutive_successes", "randomize"]
    if t.force_scale > 0.0:
        keys += ["force_rand", "pre_rb_forces", "pre_random_force_prob"]
    rec = {k: [] for k in keys}
    init = {"dof_lower": lo.numpy().copy(), "dof_upper": hi.numpy().copy(),
            "object_init_state": t.object_init_state.numpy().copy(),
            "goal_init_state": t.goal_init_state.numpy().copy(),
            "goal_states0": t.goal_states.numpy().copy(), "root0": t.root_state_tensor.numpy().copy(),
            "object_indices": t.object_indices.numpy().copy(),
            "goal_object_indices": t.goal_object_indices.numpy().copy()}
    torch.manual_seed(seed + 100)
    selected_total = 0
    for step in range(T):
        # a few envs near the end of the episode, a few with only the goal flag
        if step == 1:
            t.progress_buf[:4] = torch.tensor([596, 597, 598, 599])
        actions = torch.rand(N, D, generator=g) * 2 - 1
        if poison == "extreme":
            _scatter_extremes(g, [actions], 0.01)
        elif poison and poison != "nonunit":  # a policy that blew up in a few environments
            o = 40 + 10 * step
            actions[o, 2], actions[o + 1, 0], actions[o + 1, 5] = float("nan"), float("inf"), -float("inf")
        rec["actions"].append(actions.numpy().copy())
        env_ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        goal_ids = t.reset_goal_buf.nonzero(as_tuple=False).squeeze(-1)
        st = torch.get_rng_state()
        t.pre_physics_step(actions)
        dense = np.zeros((N, 45), np.float32)
        shapes = []
        if len(goal_ids) > 0:
            shapes.append((len(goal_ids), 4))
        if len(env_ids) > 0:
            shapes += [(len(env_ids), 37), (len(env_ids), 4), (len(env_ids),)]
        draws = _replay_uniforms(st, shapes)
        k = 0
        if len(goal_ids) > 0:
            dense[goal_ids.numpy(), 0:4] = draws[k].numpy()
            k += 1
        if len(env_ids) > 0:
            dense[env_ids.numpy(), 4:41] = draws[k].numpy()
            dense[env_ids.numpy(), 41:45] = draws[k + 1].numpy()
        rec["rand"].append(dense)
        if t.force_scale > 0.0:
            fr, nsel = _force_draws(t, N, st, shapes, env_ids)
            rec["force_rand"].append(fr)
            rec["pre_rb_forces"].append(t.rb_forces.numpy().copy())
            rec["pre_random_force_prob"].append(t.random_force_prob.numpy().copy())
            selected_total += nsel
        rec["pre_dof_state"].append(t.dof_state.numpy().copy())
        rec["pre_root"].append(t.root_state_tensor.numpy().copy())
        rec["pre_goal_states"].append(t.goal_states.numpy().copy())
        rec["pre_prev_targets"].append(t.prev_targets.numpy().copy())
        rec["pre_cur_targets"].append(t.cur_targets.numpy().copy())
        rec["pre_reset"].append(t.reset_buf.numpy().copy())
        rec["pre_reset_goal"].append(t.reset_goal_buf.numpy().copy())
        rec["pre_progress"].append(t.progress_buf.numpy().copy())
        rec["pre_successes"].append(t.successes.numpy().copy())

        # "physics": new joint state, joint forces and object state
        u = torch.rand(N, D, generator=g) * 1.04 - 0.02
        t.dof_state.view(N, D, 2)[..., 0] = lo + (hi - lo) * u
        t.dof_state.view(N, D, 2)[..., 1] = torch.randn(N, D, generator=g)
        t.dof_force_tensor = torch.randn(N, D, generator=g)
        obj = torch.zeros(N, 13)
        off = torch.randn(N, 3, generator=g)
        off = off / off.norm(dim=-1, keepdim=True) * (torch.rand(N, 1, generator=g) * 0.3)
        obj[:, 0:3] = t.goal_states[:, 0:3] + off
        q = torch.randn(N, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        # envs [8, 8+N/8): object within ~0.12 rad of the goal orientation, straddling
        # the success threshold; envs [4, 8): exactly the goal orientation
        m = N // 8
        ang = 0.08 + 0.04 * torch.rand(m, generator=g)
        ax = torch.randn(m, 3, generator=g)
        dq = tj.quat_from_angle_axis(ang, ax)
        q[8:8 + m] = tj.quat_mul(dq, t.goal_states[8:8 + m, 3:7])
        q[4:8] = t.goal_states[4:8, 3:7]
        obj[:, 3:7] = q
        obj[:, 7:13] = torch.randn(N, 6, generator=g)
        if poison == "extreme":
            _scatter_extremes(g, [obj, t.dof_state, t.dof_force_tensor], 0.01)
        elif poison == "nonunit":
            _scale_quats(g, obj[:, 3:7])  # (the goal rotations are the task's own products: unit length)
        elif poison:  # a simulation that blew up in a few other environments
            o = 40 + 10 * step
            nan, inf = float("nan"), float("inf")
            obj[o + 2, 0] = nan              # object position
            obj[o + 3, 3:7] = nan            # object rotation
            obj[o + 4, 1] = inf              # infinitely far from the goal: a fall
            obj[o + 5, 9] = -inf             # object velocity
            t.dof_state.view(N, D, 2)[o + 6, 3, 0] = nan
            t.dof_state.view(N, D, 2)[o + 7, 1, 1] = inf
            t.dof_force_tensor[o + 8, 2] = nan
            obj[o + 9, 3:7] = 0.0            # a zero quaternion
        t.root_state_tensor[t.object_indices] = obj
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        rec["in_dof_force"].append(t.dof_force_tensor.numpy().copy())
        rec["in_root"].append(t.root_state_tensor.numpy().copy())
        t.post_physics_step()
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["states"].append(t.states_buf.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["reset_goal"].append(t.reset_goal_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
        rec["successes"].append(t.successes.numpy().copy())
        rec["consecutive_successes"].append(t.consecutive_successes.numpy().copy())
        rec["randomize"].append(t.randomize_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    av = float(t.av_factor.item())
    out["meta"] = _meta(task="AllegroHand", variant=name, N=N, T=T, seed=seed, obs_type=obs_type,
                        asymmetric=asymmetric, object_type=object_type,
                        max_consecutive_successes=max_consecutive_successes,
                        use_relative_control=use_relative_control,
                        dist_reward_scale=t.dist_reward_scale, rot_reward_scale=t.rot_reward_scale,
                        rot_eps=t.rot_eps, action_penalty_scale=t.action_penalty_scale,
                        success_tolerance=tol, success_threshold_c=cstar,
                        reach_goal_bonus=float(t.reach_goal_bonus), fall_dist=t.fall_dist,
                        fall_penalty=t.fall_penalty, max_episode_length=t.max_episode_length,
                        av_factor=av, one_minus_av_factor=1.0 - av,
                        reset_position_noise=t.reset_position_noise, reset_dof_pos_noise=t.reset_dof_pos_noise,
                        reset_dof_vel_noise=t.reset_dof_vel_noise, act_moving_average=float(t.act_moving_average),
                        dof_speed_scale_dt=t.shadow_hand_dof_speed_scale * t.dt,
                        force_scale=float(t.force_scale), force_decay=float(t.force_decay), dt=float(t.dt),
                        force_decay_interval=float(t.force_decay_interval),
                        force_prob_range=[float(x) for x in force_prob_range],
                        object_rb_handle=int(t.object_rb_handles[0]), object_rb_mass=float(t.object_rb_masses[0]),
                        num_rb_force_bodies=int(t.rb_forces.shape[1]), forces_selected=selected_total)
    np.savez_compressed(os.path.join(OUT, "allegro_%s.npz" % name), **out)
    print("allegro_%s.npz" % name, "N", N, "T", T, "c*", cstar,
          "successes", int(out["reset_goal"].sum()), "resets", int(out["reset"].sum()))


def gen_allegro():
    _gen_allegro("default", N=224, T=4, seed=4)
    _gen_allegro("full", N=96, T=2, seed=5, obs_type="full")
    _gen_allegro("no_vel_asym", N=96, T=2, seed=6, obs_type="full_no_vel", asymmetric=True)
    _gen_allegro("pen_maxsucc", N=96, T=3, seed=7, object_type="pen", max_consecutive_successes=2)
    _gen_allegro("relative", N=96, T=2, seed=8, use_relative_control=True)
    _gen_allegro("ema", N=96, T=2, seed=9, act_moving_average=0.3)


def gen_hand_forces():
    """forceScale > 0: the random disturbance forces of both hand tasks (rb_forces decay, Bernoulli
    selection against random_force_prob, randn * mass * force_scale, log-uniform re-draw on reset)."""
    _gen_allegro("force", N=160, T=4, seed=10, force_scale=1.0, force_prob_range=(0.05, 0.6))
    _gen_shadow("force", N=70, T=3, seed=18, force_scale=0.5, force_prob_range=(0.02, 0.5))


def _gen_shadow(name, N, T, seed, obs_type="full_state", asymmetric=False, object_type="block",
                max_consecutive_successes=0, use_relative_control=False, act_moving_average=None,
                force_scale=None, force_prob_range=(0.001, 0.1), poison=False):
    """ShadowHand (tasks/shadow_hand.py): pre_physics_step (resets + targets of the 20 actuated joints)
    and post_physics_step (observation layouts :473-584, compute_hand_reward :747-800) of the reference
    class on synthetic simulator tensors."""
    _install_gym_stubs()
    mod = ref_shim.load("isaacgymenvs.tasks.shadow_hand")
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    cfg = _cfg("ShadowHand")
    e = cfg["env"]

    class Fake(mod.ShadowHand):
        def __init__(self):
            pass

    g = torch.Generator().manual_seed(seed)
    t = Fake()
    D, A, F, NB = 24, 20, 5, 28
    t.cfg = cfg
    t.device = "cpu"
    t.num_envs = N
    t.gym, t.sim, t.viewer, t.debug_viz = _Any(), None, None, False
    t.randomize = False
    for k_attr, k_cfg in (("dist_reward_scale", "distRewardScale"), ("rot_reward_scale", "rotRewardScale"),
                          ("action_penalty_scale", "actionPenaltyScale"), ("success_tolerance", "successTolerance"),
                          ("reach_goal_bonus", "reachGoalBonus"), ("fall_dist", "fallDistance"),
                          ("fall_penalty", "fallPenalty"), ("rot_eps", "rotEps"),
                          ("reset_position_noise", "resetPositionNoise"), ("reset_rotation_noise", "resetRotationNoise"),
                          ("reset_dof_pos_noise", "resetDofPosRandomInterval"),
                          ("reset_dof_vel_noise", "resetDofVelRandomInterval"),
                          ("shadow_hand_dof_speed_scale", "dofSpeedScale"), ("max_episode_length", "episodeLength")):
        setattr(t, k_attr, e[k_cfg])
    t.vel_obs_scale = 0.2
    t.force_torque_obs_scale = 10.0
    t.force_scale = e.get("forceScale", 0.0) if force_scale is None else force_scale
    t.use_relative_control = use_relative_control
    t.act_moving_average = e["actionsMovingAverage"] if act_moving_average is None else act_moving_average
    t.print_success_stat = False
    t.max_consecutive_successes = max_consecutive_successes
    t.av_factor = torch.tensor(e.get("averFactor", 0.1), dtype=torch.float)
    t.object_type = object_type
    t.obs_type = obs_type
    t.asymmetric_obs = asymmetric
    t.up_axis_idx = 2
    t.dt = cfg["sim"]["dt"]
    t.num_shadow_hand_dofs = D
    t.num_actions = A
    t.num_fingertips = F
    t.extras = {}
    nobs = {"openai": 42, "full_no_vel": 77, "full": 157, "full_state": 211}[obs_type]
    lo = -(torch.rand(D, generator=g) * 0.5 + 0.2)
    hi = torch.rand(D, generator=g) * 1.2 + 0.3
    t.shadow_hand_dof_lower_limits, t.shadow_hand_dof_upper_limits = lo, hi
    t.shadow_hand_dof_default_pos = torch.zeros(D)
    t.shadow_hand_dof_default_vel = torch.zeros(D)
    t.shadow_hand_default_dof_pos = torch.zeros(D)
    # the four coupled distal joints of the fingers are not actuated
    actuated = [j for j in range(D) if j not in (5, 9, 13, 18)]
    t.actuated_dof_indices = torch.tensor(actuated, dtype=torch.long)
    t.fingertip_handles = torch.tensor([7, 11, 15, 20, 25], dtype=torch.long)
    t.dof_state = torch.zeros(N * D, 2)
    t.shadow_hand_dof_state = t.dof_state.view(N, -1, 2)[:, :D]
    t.shadow_hand_dof_pos = t.shadow_hand_dof_state[..., 0]
    t.shadow_hand_dof_vel = t.shadow_hand_dof_state[..., 1]
    t.dof_force_tensor = torch.zeros(N, D)
    t.vec_sensor_tensor = torch.zeros(N, 6 * F)
    t.rigid_body_states = torch.zeros(N, NB, 13)
    t.root_state_tensor = torch.zeros(3 * N, 13)
    t.hand_indices = torch.arange(N) * 3
    t.object_indices = torch.arange(N) * 3 + 1
    t.goal_object_indices = torch.arange(N) * 3 + 2
    t.object_init_state = torch.zeros(N, 13)
    t.object_init_state[:, 0:3] = torch.tensor([0.0, -0.39, 0.6]) + 0.01 * torch.randn(N, 3, generator=g)
    t.object_init_state[:, 6] = 1.0
    t.goal_states = t.object_init_state.clone()
    t.goal_states[:, 2] -= 0.04
    t.goal_init_state = t.goal_states.clone()
    t.goal_displacement_tensor = torch.tensor([-0.2, -0.06, 0.12])
    t.prev_targets = torch.zeros(N, D)
    t.cur_targets = torch.zeros(N, D)
    t.x_unit_tensor = torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1)
    t.y_unit_tensor = torch.tensor([0.0, 1.0, 0.0]).repeat(N, 1)
    t.z_unit_tensor = torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1)
    t.obs_buf = torch.zeros(N, nobs)
    t.states_buf = torch.zeros(N, 211 if asymmetric else 0)
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.ones(N, dtype=torch.long)
    t.reset_goal_buf = t.reset_buf.clone()
    t.progress_buf = torch.zeros(N, dtype=torch.long)
    t.randomize_buf = torch.zeros(N, dtype=torch.long)
    t.successes = torch.zeros(N)
    t.consecutive_successes = torch.zeros(1)
    t.force_decay = torch.tensor(0.99)
    t.force_prob_range = torch.tensor([float(force_prob_range[0]), float(force_prob_range[1])])
    t.force_decay_interval = e.get("forceDecayInterval", 0.08)
    t.random_force_prob = torch.zeros(N)
    t.rb_forces = torch.zeros(N, NB, 3)
    t.object_rb_handles = torch.tensor([26])
    t.object_rb_masses = torch.tensor([0.1])

    tol = t.success_tolerance * (2.0 if object_type == "pen" else 1.0)
    cstar = success_threshold_c(tol)
    keys = ["actions", "rand", "pre_dof_state", "pre_root", "pre_goal_states", "pre_prev_targets", "pre_cur_targets",
            "pre_reset", "pre_reset_goal", "pre_progress", "pre_successes",
            "in_dof_state", "in_dof_force", "in_sensors", "in_bodies", "in_root", "obs", "states", "rew", "reset",
            "reset_goal", "progress", "successes", "consecutive_successes", "randomize"]
    if t.force_scale > 0.0:
        keys += ["force_rand", "pre_rb_forces", "pre_random_force_prob"]
    rec = {k: [] for k in keys}
    init = {"dof_lower": lo.numpy().copy(), "dof_upper": hi.numpy().copy(),
            "object_init_state": t.object_init_state.numpy().copy(),
            "goal_init_state": t.goal_init_state.numpy().copy(),
            "goal_states0": t.goal_states.numpy().copy(), "root0": t.root_state_tensor.numpy().copy(),
            "object_indices": t.object_indices.numpy().copy(),
            "goal_object_indices": t.goal_object_indices.numpy().copy(),
            "actuated_dof_indices": np.array(actuated, np.int32),
            "fingertip_handles": t.fingertip_handles.numpy().astype(np.int32)}
    R = 61
    torch.manual_seed(seed + 100)
    selected_total = 0
    for step in range(T):
        if step == 1:  # a few envs near the end of the episode
            L = t.max_episode_length
            t.progress_buf[:4] = torch.tensor([L - 4, L - 3, L - 2, L - 1])
        actions = torch.rand(N, A, generator=g) * 2 - 1
        if poison == "extreme":
            _scatter_extremes(g, [actions], 0.01)
        elif poison and poison != "nonunit":  # a policy that blew up in a few environments
            o = 30 + 12 * step
            actions[o, 2], actions[o + 1, 0], actions[o + 1, 7] = float("nan"), float("inf"), -float("inf")
        rec["actions"].append(actions.numpy().copy())
        env_ids = t.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        goal_ids = t.reset_goal_buf.nonzero(as_tuple=False).squeeze(-1)
        st = torch.get_rng_state()
        t.pre_physics_step(actions)
        dense = np.zeros((N, R), np.float32)
        shapes = []
        if len(goal_ids) > 0:
            shapes.append((len(goal_ids), 4))
        if len(env_ids) > 0:
            shapes += [(len(env_ids), 2 * D + 5), (len(env_ids), 4), (len(env_ids),)]
        draws = _replay_uniforms(st, shapes)
        k = 0
        if len(goal_ids) > 0:
            dense[goal_ids.numpy(), 0:4] = draws[k].numpy()
            k += 1
        if len(env_ids) > 0:
            dense[env_ids.numpy(), 4:4 + 2 * D + 5] = draws[k].numpy()
            dense[env_ids.numpy(), 4 + 2 * D + 5:R] = draws[k + 1].numpy()
        rec["rand"].append(dense)
        if t.force_scale > 0.0:
            fr, nsel = _force_draws(t, N, st, shapes, env_ids)
            rec["force_rand"].append(fr)
            rec["pre_rb_forces"].append(t.rb_forces.numpy().copy())
            rec["pre_random_force_prob"].append(t.random_force_prob.numpy().copy())
            selected_total += nsel
        rec["pre_dof_state"].append(t.dof_state.numpy().copy())
        rec["pre_root"].append(t.root_state_tensor.numpy().copy())
        rec["pre_goal_states"].append(t.goal_states.numpy().copy())
        rec["pre_prev_targets"].append(t.prev_targets.numpy().copy())
        rec["pre_cur_targets"].append(t.cur_targets.numpy().copy())
        rec["pre_reset"].append(t.reset_buf.numpy().copy())
        rec["pre_reset_goal"].append(t.reset_goal_buf.numpy().copy())
        rec["pre_progress"].append(t.progress_buf.numpy().copy())
        rec["pre_successes"].append(t.successes.numpy().copy())

        # "physics": new joint state, joint forces, sensors, body states and object state
        u = torch.rand(N, D, generator=g) * 1.04 - 0.02
        t.dof_state.view(N, D, 2)[..., 0] = lo + (hi - lo) * u
        t.dof_state.view(N, D, 2)[..., 1] = torch.randn(N, D, generator=g)
        t.dof_force_tensor = torch.randn(N, D, generator=g)
        t.vec_sensor_tensor = torch.randn(N, 6 * F, generator=g)
        t.rigid_body_states = torch.randn(N, NB, 13, generator=g)
        obj = torch.zeros(N, 13)
        off = torch.randn(N, 3, generator=g)
        off = off / off.norm(dim=-1, keepdim=True) * (torch.rand(N, 1, generator=g) * 0.3)
        obj[:, 0:3] = t.goal_states[:, 0:3] + off
        q = torch.randn(N, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        # envs [8, 8+N/8): object within ~0.12 rad of the goal orientation, straddling the success
        # threshold; envs [4, 8): exactly the goal orientation
        m = N // 8
        ang = 0.08 + 0.04 * torch.rand(m, generator=g) * (2.0 if object_type == "pen" else 1.0)
        ax = torch.randn(m, 3, generator=g)
        dq = tj.quat_from_angle_axis(ang, ax)
        q[8:8 + m] = tj.quat_mul(dq, t.goal_states[8:8 + m, 3:7])
        q[4:8] = t.goal_states[4:8, 3:7]
        obj[:, 3:7] = q
        obj[:, 7:13] = torch.randn(N, 6, generator=g)
        if poison == "extreme":
            _scatter_extremes(g, [obj, t.dof_state, t.dof_force_tensor], 0.01)
        elif poison == "nonunit":
            _scale_quats(g, obj[:, 3:7])  # (the goal rotations are the task's own products: unit length)
        elif poison:  # a simulation that blew up in a few other environments
            o = 30 + 12 * step
            nan, inf = float("nan"), float("inf")
            obj[o + 2, 0] = nan
            obj[o + 3, 3:7] = nan
            obj[o + 4, 1] = inf
            obj[o + 5, 9] = -inf
            t.dof_state.view(N, D, 2)[o + 6, 3, 0] = nan
            t.dof_state.view(N, D, 2)[o + 7, 1, 1] = inf
            t.dof_force_tensor[o + 8, 2] = nan
            obj[o + 9, 3:7] = 0.0
            t.vec_sensor_tensor[o + 10, 4] = nan
            t.rigid_body_states[o + 11, :, 0:7] = nan     # every body of one environment (the fingertips among them)
        t.root_state_tensor[t.object_indices] = obj
        rec["in_dof_state"].append(t.dof_state.numpy().copy())
        rec["in_dof_force"].append(t.dof_force_tensor.numpy().copy())
        rec["in_sensors"].append(t.vec_sensor_tensor.numpy().copy())
        rec["in_bodies"].append(t.rigid_body_states.numpy().copy())
        rec["in_root"].append(t.root_state_tensor.numpy().copy())
        t.post_physics_step()
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["states"].append(t.states_buf.numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["reset_goal"].append(t.reset_goal_buf.numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
        rec["successes"].append(t.successes.numpy().copy())
        rec["consecutive_successes"].append(t.consecutive_successes.numpy().copy())
        rec["randomize"].append(t.randomize_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(init)
    av = float(t.av_factor.item())
    out["meta"] = _meta(task="ShadowHand", variant=name, N=N, T=T, seed=seed, obs_type=obs_type,
                        asymmetric=asymmetric, object_type=object_type, num_bodies=NB,
                        max_consecutive_successes=max_consecutive_successes,
                        use_relative_control=use_relative_control,
                        dist_reward_scale=t.dist_reward_scale, rot_reward_scale=t.rot_reward_scale,
                        rot_eps=t.rot_eps, action_penalty_scale=t.action_penalty_scale,
                        success_tolerance=tol, success_threshold_c=cstar,
                        reach_goal_bonus=float(t.reach_goal_bonus), fall_dist=t.fall_dist,
                        fall_penalty=t.fall_penalty, max_episode_length=t.max_episode_length,
                        av_factor=av, one_minus_av_factor=1.0 - av,
                        reset_position_noise=t.reset_position_noise, reset_dof_pos_noise=t.reset_dof_pos_noise,
                        reset_dof_vel_noise=t.reset_dof_vel_noise, act_moving_average=float(t.act_moving_average),
                        dof_speed_scale_dt=t.shadow_hand_dof_speed_scale * t.dt,
                        force_scale=float(t.force_scale), force_decay=float(t.force_decay), dt=float(t.dt),
                        force_decay_interval=float(t.force_decay_interval),
                        force_prob_range=[float(x) for x in force_prob_range],
                        object_rb_handle=int(t.object_rb_handles[0]), object_rb_mass=float(t.object_rb_masses[0]),
                        num_rb_force_bodies=int(t.rb_forces.shape[1]), forces_selected=selected_total)
    np.savez_compressed(os.path.join(OUT, "shadow_%s.npz" % name), **out)
    print("shadow_%s.npz" % name, "N", N, "T", T, "c*", cstar,
          "successes", int(out["reset_goal"].sum()), "resets", int(out["reset"].sum()))


def gen_shadow():
    _gen_shadow("default", N=70, T=3, seed=14)                       # full_state, 211 columns
    _gen_shadow("openai_asym", N=70, T=2, seed=15, obs_type="openai", asymmetric=True)
    _gen_shadow("full", N=70, T=2, seed=16, obs_type="full", use_relative_control=True)
    _gen_shadow("no_vel_pen", N=70, T=3, seed=17, obs_type="full_no_vel", object_type="pen",
                max_consecutive_successes=2, act_moving_average=0.3)


# --------------------------------------------------------------------------
AMP_NODE_NAMES = ["pelvis", "torso", "head", "right_upper_arm", "right_lower_arm", "right_hand",
                  "left_upper_arm", "left_lower_arm", "left_hand", "right_thigh", "right_shin", "right_foot",
                  "left_thigh", "left_shin", "left_foot"]
AMP_PARENTS = [-1, 0, 1, 1, 3, 4, 1, 6, 7, 0, 9, 10, 0, 12, 13]
# plausible link offsets of a 15-body humanoid (the reference's t-pose file is a
# pickle and is deliberately not loaded; the offsets only need to be non-trivial)
AMP_LOCAL_T = [[0, 0, 0], [0, 0, 0.236], [0, 0, 0.224], [-0.024, -0.183, 0.243], [0, 0, -0.275], [0, 0, -0.259],
               [-0.024, 0.183, 0.243], [0, 0, -0.275], [0, 0, -0.259], [0, -0.085, 0], [0, 0, -0.421],
               [0, 0, -0.41], [0, 0.085, 0], [0, 0, -0.421], [0, 0, -0.41]]


def _synth_motion(sk, tree, F, fps, seed):
    """smooth random joint rotations + a wandering root, as a reference SkeletonMotion"""
    rot3d = ref_shim.load("isaacgymenvs.tasks.amp.poselib.poselib.core.rotation3d")
    g = torch.Generator().manual_seed(seed)
    J = 15
    tt = torch.arange(F, dtype=torch.float32).unsqueeze(-1) / fps
    axis = torch.randn(J, 3, generator=g)
    axis[[4, 7, 10, 13]] = torch.tensor([0.0, 1.0, 0.0])  # elbows / knees: hinge about y
    axis = axis / axis.norm(dim=-1, keepdim=True)
    amp = torch.rand(J, generator=g) * 1.2 + 0.1
    w = torch.rand(J, generator=g) * 4 + 1
    ph = torch.rand(J, generator=g) * 6.28
    ang = amp * torch.sin(w * tt + ph)                    # [F, J]
    half = ang.unsqueeze(-1) * 0.5
    r = torch.cat([axis.unsqueeze(0) * torch.sin(half), torch.cos(half)], dim=-1)  # [F, J, 4]
    r[0, 2] = torch.tensor([0.0, 0.0, 0.0, 1.0])          # an exact identity joint on frame 0
    r = rot3d.quat_normalize(r)
    root_t = torch.stack([0.8 * tt[:, 0], 0.3 * torch.sin(1.3 * tt[:, 0]), 0.9 + 0.05 * torch.sin(3 * tt[:, 0])], dim=-1)
    state = sk.SkeletonState.from_rotation_and_root_translation(tree, r, root_t, is_local=True)
    return sk.SkeletonMotion.from_skeleton_state(state, fps=fps)


def gen_motion(seed=11):
    _install_gym_stubs()
    import tempfile
    sk = ref_shim.load("isaacgymenvs.tasks.amp.poselib.poselib.skeleton.skeleton3d")
    ml = ref_shim.load("isaacgymenvs.tasks.amp.utils_amp.motion_lib")
    hamp = ref_shim.load("isaacgymenvs.tasks.humanoid_amp")
    tree = sk.SkeletonTree(AMP_NODE_NAMES, torch.tensor(AMP_PARENTS), torch.tensor(AMP_LOCAL_T, dtype=torch.float32))
    specs = [(90, 30, 1.0), (61, 60, 3.0)]  # (frames, fps, weight)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        entries = []
        for i, (F, fps, wgt) in enumerate(specs):
            m = _synth_motion(sk, tree, F, fps, seed + i)
            m.to_file(os.path.join(d, "m%d.npy" % i))
            entries.append({"file": "m%d.npy" % i, "weight": wgt})
            out["m%d_rotation" % i] = m.rotation.numpy().copy()
            out["m%d_root_translation" % i] = m.root_translation.numpy().copy()
            out["m%d_global_velocity" % i] = m.global_velocity.numpy().copy()
            out["m%d_global_angular_velocity" % i] = m.global_angular_velocity.numpy().copy()
        with open(os.path.join(d, "set.yaml"), "w") as f:
            yaml.safe_dump({"motions": entries}, f)
        cwd = os.getcwd()
        os.chdir("/")
        lib = ml.MotionLib(motion_file=os.path.join(d, "set.yaml"), num_dofs=28,
                           key_body_ids=np.array([5, 8, 11, 14]), device="cpu")
        os.chdir(cwd)
    for i, m in enumerate(lib._motions):
        out["m%d_global_translation" % i] = m.global_translation.numpy().copy()
        out["m%d_global_rotation" % i] = m.global_rotation.numpy().copy()
        out["m%d_local_rotation" % i] = m.local_rotation.numpy().copy()
        out["m%d_dof_vels" % i] = np.asarray(m.dof_vels, np.float64).copy()
    out["parent_indices"] = np.array(AMP_PARENTS, np.int64)
    out["local_translation"] = np.array(AMP_LOCAL_T, np.float32)
    out["motion_lengths"] = lib._motion_lengths.copy()
    out["motion_weights"] = lib._motion_weights.copy()
    out["motion_dt"] = lib._motion_dt.copy()
    out["motion_num_frames"] = lib._motion_num_frames.copy()
    out["motion_fps"] = np.asarray(lib._motion_fps, np.float64).copy()

    # sampling is host numpy RNG: pin it
    np.random.seed(1234)
    ids = lib.sample_motions(600)
    times = lib.sample_time(ids)
    out["sampled_ids"], out["sampled_times"] = ids.copy(), times.copy()
    # edge cases: t = 0, t = len, beyond both ends, exact frame boundaries, history steps before the clip start
    extra_ids = np.array([0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1], np.int64)
    L = lib._motion_lengths
    extra_t = np.array([0.0, L[0], -0.05, L[0] + 0.5, 0.0, L[1], -1e-9, 10.0, 3.0 / 30.0, 7.0 / 60.0,
                        L[0] - 1e-12, 1.0 / 60.0 - 1e-15])
    ids = np.concatenate([ids, extra_ids])
    times = np.concatenate([times, extra_t])
    out["ids"], out["times"] = ids, times
    i0, i1, blend = lib._calc_frame_blend(times, lib._motion_lengths[ids], lib._motion_num_frames[ids],
                                          lib._motion_dt[ids])
    out["frame_idx0"], out["frame_idx1"], out["blend"] = i0.astype(np.int64), i1.astype(np.int64), blend
    names = ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos"]
    res = lib.get_motion_state(ids, times)
    for k, v in zip(names, res):
        out[k] = v.numpy().copy()
    root_states = torch.cat([res[0], res[1], res[3], res[4]], dim=-1)
    for lro in (False, True):
        out["amp_obs_local%d" % int(lro)] = hamp.build_amp_observations(root_states, res[2], res[5], res[6], lro).numpy()
    out["meta"] = _meta(task="MotionLib", seed=seed, specs=specs, np_seed=1234, key_body_ids=[5, 8, 11, 14])
    np.savez_compressed(os.path.join(OUT, "motion.npz"), **out)
    print("motion.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
    _gen_amp_task(lib, hamp, seed)
    # every switch the other way: root observation in the heading frame, a deeper history, other thresholds
    _gen_amp_task(lib, hamp, seed + 7, N=64, K=3, name="amp_alt",
                  over=dict(localRootObs=True, terminationHeight=0.62, episodeLength=150))
    _gen_amp_task(lib, hamp, seed + 9, N=64, K=2, name="amp_nonfinite", poison=True)
    _gen_amp_task(lib, hamp, seed + 13, N=64, K=2, T=3, name="amp_nonunit", over=dict(localRootObs=True), poison="nonunit")
    _gen_amp_task(lib, hamp, seed + 11, N=96, K=2, T=3, name="amp_extreme", over=dict(localRootObs=True), poison="extreme")  # N=96: the file stays below 1 MB


def _gen_amp_task(lib, hamp, seed, N=96, K=2, T=4, name="amp", over=None, poison=False):
    """HumanoidAMP.post_physics_step / reset_idx of the reference task class on
    synthetic simulator tensors, with the reference MotionLib built above."""
    cfg = _cfg("HumanoidAMP")
    e = cfg["env"]
    e.update(over or {})
    g = torch.Generator().manual_seed(seed + 50)

    class Fake(hamp.HumanoidAMP):
        def __init__(self):
            pass

    t = Fake()
    t.cfg = cfg
    t.device = "cpu"
    t.num_envs = N
    t.num_dof, t.num_bodies = 28, 15
    t.gym, t.sim, t.viewer, t.debug_viz = _Any(), None, None, False
    t.max_episode_length = e["episodeLength"]
    t._local_root_obs = e["localRootObs"]
    t._termination_height = e["terminationHeight"]
    t._enable_early_termination = e["enableEarlyTermination"]
    t._state_init = hamp.HumanoidAMP.StateInit[e["stateInit"]]
    t._hybrid_init_prob = e["hybridInitProb"]
    t._num_amp_obs_steps = K
    t.dt = e["controlFrequencyInv"] * cfg["sim"]["dt"]
    t._reset_default_env_ids, t._reset_ref_env_ids = [], []
    t._motion_lib = lib
    t._key_body_ids = torch.tensor([5, 8, 11, 14])
    t._contact_body_ids = torch.tensor([11, 14])
    t._root_states = torch.zeros(N, 13)
    t._initial_root_states = torch.zeros(N, 13)
    t._dof_state = torch.zeros(N * 28, 2)
    t._dof_pos = t._dof_state.view(N, 28, 2)[..., 0]
    t._dof_vel = t._dof_state.view(N, 28, 2)[..., 1]
    t._initial_dof_pos = torch.zeros(N, 28)
    t._initial_dof_vel = torch.zeros(N, 28)
    t._rigid_body_state = torch.zeros(N * 15, 13)
    t._rigid_body_pos = t._rigid_body_state.view(N, 15, 13)[..., 0:3]
    t._contact_forces = torch.zeros(N, 15, 3)
    t._terminate_buf = torch.ones(N, dtype=torch.long)
    t.obs_buf = torch.zeros(N, 105)
    t.rew_buf = torch.zeros(N)
    t.reset_buf = torch.ones(N, dtype=torch.long)
    t.progress_buf = torch.zeros(N, dtype=torch.long)
    t.extras = {}
    t.num_amp_obs = K * 105
    t.actions = torch.zeros(N, 28)
    t._amp_obs_buf = torch.zeros(N, K, 105)
    t._curr_amp_obs_buf = t._amp_obs_buf[:, 0]
    t._hist_amp_obs_buf = t._amp_obs_buf[:, 1:]

    def physics():
        root = torch.zeros(N, 13)
        root[:, 0:3] = torch.randn(N, 3, generator=g) * 0.3
        root[:, 2] = 0.9 + 0.1 * torch.randn(N, generator=g)
        q = torch.randn(N, 4, generator=g)
        root[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
        root[0, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0])
        root[:, 7:13] = torch.randn(N, 6, generator=g)
        t._root_states.copy_(root)
        ds = torch.randn(N, 28, 2, generator=g)
        ds[:, :, 0] *= 1.2
        ds[1, 0:3, 0] = 0.0          # zero exponential map: the masked 0/0 lanes
        ds[2, 0:3, 0] = torch.tensor([1e-7, 0.0, 0.0])
        ds[3, 0:3, 0] = torch.tensor([3.5, 1.0, 0.5])   # angle > pi
        t._dof_state.copy_(ds.view(N * 28, 2))
        rb = torch.randn(N, 15, 13, generator=g)
        rb[:, :, 0:3] = root[:, None, 0:3] + 0.4 * torch.randn(N, 15, 3, generator=g)
        rb[:, :, 2] = 0.45 + 0.6 * torch.rand(N, 15, generator=g)  # a few bodies below the 0.5 m threshold
        t._rigid_body_state.copy_(rb.view(N * 15, 13))
        cf = torch.randn(N, 15, 3, generator=g) * 0.03          # below the 0.1 threshold ...
        hit = torch.rand(N, generator=g) < 0.3                  # ... except one body of ~30 % of the envs
        body = torch.randint(0, 15, (N,), generator=g)
        cf[hit, body[hit], 0] = 0.5
        cf[:, [11, 14]] = 5.0                                   # feet always in contact
        cf[4] = -3.0                                            # negative components never count
        cf[5, 3, 1] = 0.1                                       # exactly 0.1: not > 0.1
        t._contact_forces.copy_(cf)
        if poison == "extreme":
            _scatter_extremes(g, [t._root_states, t._dof_state, t._rigid_body_state, t._contact_forces], 0.004)
        elif poison == "nonunit":
            _scale_quats(g, t._root_states[:, 3:7])
        elif poison:  # non-finite simulator state in a few environments
            nan, inf = float("nan"), float("inf")
            t._root_states[40, 0], t._root_states[41, 6], t._root_states[42, 2] = nan, nan, inf
            t._root_states[43, 8], t._root_states[44, 3:7] = -inf, 0.0
            t._dof_state.view(N, 28, 2)[45, 1, 0] = nan           # inside a 3-dof joint
            t._dof_state.view(N, 28, 2)[46, 9, 0] = inf           # a hinge
            t._dof_state.view(N, 28, 2)[47, 4, 1] = nan           # a joint velocity
            t._rigid_body_state.view(N, 15, 13)[48, 3, 2] = nan   # a body height: not < the termination height
            t._rigid_body_state.view(N, 15, 13)[49, 3, 2] = -inf
            t._rigid_body_state.view(N, 15, 13)[50, 5, 0] = nan   # a key body
            t._contact_forces[51, 2, 0], t._contact_forces[52, 2, 1] = nan, inf
            t._contact_forces[49, 2, 1] = inf                     # together with the body at -inf: a fall

    keys = ["in_root", "in_dof_state", "in_rigid_body_state", "in_contact_forces", "obs", "amp_obs", "rew", "reset",
            "terminate", "progress"]
    rec = {k: [] for k in keys}
    rst = {}
    for step in range(T):
        if step == 1:
            t.progress_buf[:4] = torch.tensor([296, 297, 298, 299])
        if step == 2:
            # reset_done(): reset_idx on the flagged envs, Random state init (motion library)
            env_ids = t.reset_buf.nonzero(as_tuple=False).flatten()
            rst["env_ids"] = env_ids.numpy().copy()
            np.random.seed(777)
            t.reset_idx(env_ids)
            rst["motion_ids"] = np.asarray(t._reset_ref_motion_ids).copy()
            rst["motion_times"] = np.asarray(t._reset_ref_motion_times).copy()
            for k, v in (("root", t._root_states), ("dof_state", t._dof_state), ("obs", t.obs_buf),
                         ("amp_obs", t._amp_obs_buf), ("progress", t.progress_buf), ("reset", t.reset_buf),
                         ("terminate", t._terminate_buf)):
                rst["after_" + k] = v.numpy().copy()
        physics()
        rec["in_root"].append(t._root_states.numpy().copy())
        rec["in_dof_state"].append(t._dof_state.numpy().copy())
        rec["in_rigid_body_state"].append(t._rigid_body_state.numpy().copy())
        rec["in_contact_forces"].append(t._contact_forces.numpy().copy())
        t.post_physics_step()
        rec["obs"].append(t.obs_buf.numpy().copy())
        rec["amp_obs"].append(t.extras["amp_obs"].numpy().copy())
        rec["rew"].append(t.rew_buf.numpy().copy())
        rec["reset"].append(t.reset_buf.numpy().copy())
        rec["terminate"].append(t.extras["terminate"].numpy().copy())
        rec["progress"].append(t.progress_buf.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update({"reset_" + k: v for k, v in rst.items()})
    out["meta"] = _meta(task="HumanoidAMP", N=N, K=K, T=T, seed=seed, dt=t.dt, local_root_obs=bool(t._local_root_obs),
                        termination_height=t._termination_height, max_episode_length=t.max_episode_length,
                        enable_early_termination=bool(t._enable_early_termination), reset_before_step=2,
                        np_seed=777, state_init=e["stateInit"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")},
          "terminated", int(out["terminate"].sum()), "resets", int(out["reset"].sum()))


def gen_vectask(seed=21, N=301, T=3):
    """VecTask.allocate_buffers / step tail / reset (tasks/base/vec_task.py:310-333,
    :401-418, :436-449) of the reference's real base class, two configurations."""
    vt = ref_shim.load_real_vec_task()
    g = torch.Generator().manual_seed(seed)
    out = {}
    variants = {"clip5": dict(clip=5.0, num_obs=13, num_states=7, max_len=50),
                "noclip": dict(clip=float("inf"), num_obs=60, num_states=0, max_len=1000)}
    for name, v in variants.items():
        frames = []

        class Task(vt.VecTask):
            def __init__(self):
                pass

            def pre_physics_step(self, actions):
                self.actions = actions.clone()

            def post_physics_step(self):
                f = frames[self.control_steps]
                self.obs_buf[:] = f["obs"]
                if self.num_states:
                    self.states_buf[:] = f["states"]
                self.rew_buf[:] = f["rew"]
                self.reset_buf[:] = f["reset"]
                self.progress_buf[:] = f["progress"]

        t = Task()
        t.num_environments, t.num_observations, t.num_states, t.num_actions = N, v["num_obs"], v["num_states"], 3
        t.device = t.rl_device = "cpu"
        t.clip_obs, t.clip_actions = v["clip"], 1.0
        t.dr_randomizations, t.control_freq_inv, t.force_render = {}, 1, False
        t.gym, t.sim, t.control_steps, t.obs_dict = _Any(), None, 0, {}
        t.max_episode_length = v["max_len"]
        t.allocate_buffers()
        for k in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "timeout_buf", "progress_buf", "randomize_buf"):
            out["%s_alloc_%s" % (name, k)] = getattr(t, k).numpy().copy()
        first = t.reset()
        out[name + "_reset_obs"] = first["obs"].numpy().copy()
        for step in range(T):
            obs = torch.randn(N, v["num_obs"], generator=g) * 4.0
            obs[0, 0], obs[1, 1], obs[2, 2] = float("nan"), float("inf"), float("-inf")
            obs[3, 0], obs[4, 0] = 5.0, -5.0
            progress = torch.randint(0, v["max_len"] + 2, (N,), generator=g)
            progress[:4] = torch.tensor([v["max_len"] - 2, v["max_len"] - 1, v["max_len"], v["max_len"] - 1])
            reset = (torch.rand(N, generator=g) < 0.5).long()
            reset[:4] = torch.tensor([1, 1, 1, 0])
            frames.append(dict(obs=obs, states=torch.randn(N, max(v["num_states"], 1), generator=g)[:, :v["num_states"]] * 6.0,
                               rew=torch.randn(N, generator=g), reset=reset, progress=progress))
        acts, outs = [], dict(obs=[], states=[], rew=[], reset=[], time_outs=[])
        for step in range(T):
            a = torch.randn(N, 3, generator=g) * 2.0
            acts.append(a.numpy().copy())
            obs_dict, rew, reset, extras = t.step(a)
            outs["obs"].append(obs_dict["obs"].numpy().copy())
            if v["num_states"]:
                outs["states"].append(obs_dict["states"].numpy().copy())
            outs["rew"].append(rew.numpy().copy())
            outs["reset"].append(reset.numpy().copy())
            outs["time_outs"].append(extras["time_outs"].numpy().copy())
            out.setdefault(name + "_clipped_actions", []).append(t.actions.numpy().copy())
        for k in ("obs", "states", "rew", "reset", "progress"):
            if k == "states" and not v["num_states"]:
                continue
            out["%s_in_%s" % (name, k)] = np.stack([f[k].numpy() for f in frames])
        out[name + "_in_actions"] = np.stack(acts)
        out[name + "_clipped_actions"] = np.stack(out[name + "_clipped_actions"])
        for k, lst in outs.items():
            if lst:
                out["%s_out_%s" % (name, k)] = np.stack(lst)
    out["meta"] = _meta(task="VecTask", seed=seed, N=N, T=T,
                        variants={k: dict(v, clip=(None if math.isinf(v["clip"]) else v["clip"])) for k, v in variants.items()})
    np.savez_compressed(os.path.join(OUT, "vectask.npz"), **out)
    print("vectask.npz", {k: (v.shape, v.dtype) for k, v in out.items() if hasattr(v, "shape")})


def gen_prestep(seed=61, N=127, T=4):
    """What surrounds the physics step on the action and observation side: the tasks'
    pre_physics_step scaling (tasks/ant.py:368-374, tasks/humanoid.py:392-396,
    tasks/amp/humanoid_amp_base.py:410-422 / :467-469, tasks/cartpole.py:164-168) and the
    observation / action noise of the real VecTask.apply_randomizations + step
    (tasks/base/vec_task.py:381-384, :407-412, :630-729)."""
    import contextlib
    import copy
    _install_gym_stubs()
    sys.modules["isaacgym.gymtorch"].unwrap_tensor = lambda t: t
    g = torch.Generator().manual_seed(seed)
    out = {}

    class Gym:  # stands where the simulator stands; keeps what the task hands over
        def __init__(self):
            self.sent, self.frame = {}, 0

        def get_frame_count(self, sim):
            return self.frame

        def set_dof_actuation_force_tensor(self, sim, t):
            self.sent["force"] = t.clone()

        def set_dof_position_target_tensor(self, sim, t):
            self.sent["pd"] = t.clone()

        def __getattr__(self, k):  # property getters / setters of the physical randomisation: unused
            return _Any()

    def fake(cls, **kw):
        t = cls.__new__(cls)
        t.device, t.gym, t.sim = "cpu", Gym(), None
        for k, v in kw.items():
            setattr(t, k, v)
        return t

    # ---- action scaling of the four task classes ----
    ant = ref_shim.load("isaacgymenvs.tasks.ant").Ant
    hum = ref_shim.load("isaacgymenvs.tasks.humanoid").Humanoid
    amp = ref_shim.load("isaacgymenvs.tasks.amp.humanoid_amp_base").HumanoidAMPBase
    cart = ref_shim.load("isaacgymenvs.tasks.cartpole").Cartpole
    a8 = torch.randn(N, 8, generator=g) * 1.5
    gears = torch.rand(8, generator=g) * 135.0 + 15.0
    for tag, ps in (("ant", _cfg("Ant")["env"]["powerScale"]), ("ant_ps037", 0.37)):
        t = fake(ant, joint_gears=gears, power_scale=ps)
        t.pre_physics_step(a8)
        out.update({tag + "_actions": a8.numpy().copy(), tag + "_gain": gears.numpy().copy(),
                    tag + "_forces": t.gym.sent["force"].numpy().copy(), tag + "_self_actions": t.actions.numpy().copy()})
        out[tag + "_power_scale"] = np.float64(ps)
    a21 = torch.randn(N, 21, generator=g)
    efforts = torch.rand(21, generator=g) * 180.0 + 20.0
    t = fake(hum, motor_efforts=efforts, power_scale=_cfg("Humanoid")["env"]["powerScale"])
    t.pre_physics_step(a21)
    out.update(humanoid_actions=a21.numpy().copy(), humanoid_gain=efforts.numpy().copy(),
               humanoid_forces=t.gym.sent["force"].numpy().copy(), humanoid_power_scale=np.float64(t.power_scale))
    a28 = torch.randn(N, 28, generator=g)
    off28, sc28 = torch.randn(28, generator=g) * 0.4, torch.rand(28, generator=g) * 2.0 + 0.3
    eff28 = torch.rand(28, generator=g) * 180.0 + 20.0
    for pd in (True, False):
        t = fake(amp, _pd_control=pd, _pd_action_offset=off28, _pd_action_scale=sc28, motor_efforts=eff28,
                 power_scale=_cfg("HumanoidAMP")["env"]["powerScale"])
        t.pre_physics_step(a28)
        out["amp_pd_targets" if pd else "amp_forces"] = t.gym.sent["pd" if pd else "force"].numpy().copy()
    out.update(amp_actions=a28.numpy().copy(), amp_offset=off28.numpy().copy(), amp_scale=sc28.numpy().copy(),
               amp_efforts=eff28.numpy().copy(), amp_power_scale=np.float64(t.power_scale))
    a1 = torch.randn(N, 1, generator=g)
    t = fake(cart, num_envs=N, num_dof=2, max_push_effort=_cfg("Cartpole")["env"]["maxEffort"])
    t.pre_physics_step(a1)
    out.update(cartpole_actions=a1.numpy().copy(), cartpole_forces=t.gym.sent["force"].numpy().copy(),
               cartpole_max_push=np.float64(t.max_push_effort))

    # ---- observation / action noise through the real VecTask ----
    vt = ref_shim.load_real_vec_task()
    variants = {
        "v1": dict(frequency=3, frames=[25, 26, 27, 28], num_obs=13, clip_obs=5.0, clip_actions=1.0,
                   observations=dict(range=[0.01, 0.3], range_correlated=[0.02, 0.1], operation="additive",
                                     distribution="gaussian", schedule="linear", schedule_steps=40),
                   actions=dict(range=[0.8, 1.2], range_correlated=[0.95, 1.05], operation="scaling",
                                distribution="uniform", schedule="constant", schedule_steps=10)),
        "v2": dict(frequency=2, frames=[3, 4, 5, 6], num_obs=60, clip_obs=float("inf"), clip_actions=0.9,
                   observations=dict(range=[0.9, 1.1], range_correlated=[0.97, 1.02], operation="scaling",
                                     distribution="uniform"),
                   actions=dict(range=[0.0, 0.1], range_correlated=[0.01, 0.03], operation="additive",
                                distribution="gaussian", schedule="linear", schedule_steps=8)),
    }
    draws = []
    real_randn, real_rand = torch.randn_like, torch.rand_like

    @contextlib.contextmanager
    def record_draws():
        def randn_like(t, *a, **k):
            r = real_randn(t, *a, **k)
            draws.append(("randn", r.numpy().copy()))
            return r

        def rand_like(t, *a, **k):
            r = real_rand(t, *a, **k)
            draws.append(("rand", r.numpy().copy()))
            return r
        torch.randn_like, torch.rand_like = randn_like, rand_like
        try:
            yield
        finally:
            torch.randn_like, torch.rand_like = real_randn, real_rand

    for name, v in variants.items():
        frames = []

        class Task(vt.VecTask):
            def __init__(self):
                pass

            def pre_physics_step(self, actions):
                self.actions = actions.clone()

            def post_physics_step(self):
                f = frames[self.control_steps]
                self.obs_buf[:] = f["obs"]
                self.reset_buf[:] = f["reset"]
                self.progress_buf[:] = f["progress"]

        t = Task()
        t.num_environments, t.num_observations, t.num_states, t.num_actions = N, v["num_obs"], 0, 5
        t.device = t.rl_device = "cpu"
        t.clip_obs, t.clip_actions = v["clip_obs"], v["clip_actions"]
        t.dr_randomizations, t.control_freq_inv, t.force_render = {}, 1, False
        t.gym, t.sim, t.control_steps, t.obs_dict = Gym(), None, 0, {}
        t.max_episode_length = 50
        t.first_randomization, t.last_rand_step, t.last_step = True, -1, -1
        t.actor_params_generator, t.envs, t.original_props, t.sim_initialized = None, [], {}, True
        t.allocate_buffers()
        dr = dict(frequency=v["frequency"], observations=v["observations"], actions=v["actions"], actor_params={})
        rec = {k: [] for k in ("in_actions", "in_obs", "in_reset", "in_progress", "self_actions", "obs_buf_after",
                               "out_obs", "time_outs", "act_corr", "act_fresh", "obs_corr", "obs_fresh",
                               "act_params", "obs_params", "randomize_before", "randomize_after", "did_randomize")}
        torch.manual_seed(seed + len(name))
        for step in range(T):
            obs = torch.randn(N, v["num_obs"], generator=g) * 3.0
            frames.append(dict(obs=obs, reset=(torch.rand(N, generator=g) < 0.4).long(),
                               progress=torch.randint(0, 52, (N,), generator=g)))
            # the tasks call apply_randomizations from reset_idx / pre_physics_step, i.e. before the step
            t.gym.frame = v["frames"][step]
            t.randomize_buf[:] = torch.randint(0, 5, (N,), generator=g)
            rec["randomize_before"].append(t.randomize_buf.numpy().copy())
            before = t.last_rand_step
            t.apply_randomizations(copy.deepcopy(dr))
            rec["did_randomize"].append(int(t.last_rand_step != before))
            rec["randomize_after"].append(t.randomize_buf.numpy().copy())

            def params(d):
                keys = ("mu", "var", "mu_corr", "var_corr") if "mu" in d else ("lo", "hi", "lo_corr", "hi_corr")
                return np.array([d[k] for k in keys], np.float64)
            rec["act_params"].append(params(t.dr_randomizations["actions"]))
            rec["obs_params"].append(params(t.dr_randomizations["observations"]))
            a = torch.randn(N, 5, generator=g)
            del draws[:]
            with record_draws():
                obs_dict, rew, reset, extras = t.step(a)
            kinds = [k for k, _ in draws]
            # per lambda: an optional new 'corr' draw (randn) followed by this call's draw
            act_fresh_kind = "rand" if v["actions"]["distribution"] == "uniform" else "randn"
            obs_fresh_kind = "rand" if v["observations"]["distribution"] == "uniform" else "randn"
            i = 0
            new_corr = len(draws) == 4
            assert len(draws) in (2, 4), kinds
            if new_corr:
                act_corr = draws[i][1]; i += 1
            act_fresh = draws[i][1]; assert draws[i][0] == act_fresh_kind; i += 1
            if new_corr:
                obs_corr = draws[i][1]; i += 1
            obs_fresh = draws[i][1]; assert draws[i][0] == obs_fresh_kind
            rec["act_corr"].append(act_corr); rec["act_fresh"].append(act_fresh)
            rec["obs_corr"].append(obs_corr); rec["obs_fresh"].append(obs_fresh)
            rec["in_actions"].append(a.numpy().copy()); rec["in_obs"].append(obs.numpy().copy())
            rec["in_reset"].append(frames[-1]["reset"].numpy().copy())
            rec["in_progress"].append(frames[-1]["progress"].numpy().copy())
            rec["self_actions"].append(t.actions.numpy().copy())
            rec["obs_buf_after"].append(t.obs_buf.numpy().copy())
            rec["out_obs"].append(obs_dict["obs"].numpy().copy())
            rec["time_outs"].append(extras["time_outs"].numpy().copy())
        for k, lst in rec.items():
            out["%s_%s" % (name, k)] = np.stack([np.asarray(x) for x in lst])
    meta_variants = {k: dict(v, clip_obs=(None if math.isinf(v["clip_obs"]) else v["clip_obs"])) for k, v in variants.items()}
    out["meta"] = _meta(task="prestep", seed=seed, N=N, T=T, variants=meta_variants, max_episode_length=50)
    np.savez_compressed(os.path.join(OUT, "prestep.npz"), **out)
    print("prestep.npz", {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, "shape")})


def gen_episode_stats(seed=31, N=777, T=6):
    """The reward / length bookkeeping at learning/common_agent.py:284-296.  That module imports
    rl_games (absent), so the expressions of the call site are evaluated here with torch on
    CPU, statement by statement: current_rewards += rewards; current_lengths += 1;
    done_indices = dones.nonzero(); update(current_rewards[done_indices]) (sum and count recorded);
    not_dones = 1.0 - dones.float(); current_rewards *= not_dones.unsqueeze(1); current_lengths *= not_dones."""
    g = torch.Generator().manual_seed(seed)
    current_rewards = torch.zeros(N, 1)
    current_lengths = torch.zeros(N)
    out = {k: [] for k in ("rew", "dones", "current_rewards", "current_lengths", "n_done", "mean_reward", "mean_length",
                           "sum_reward", "sum_length")}
    for t in range(T):
        rewards = torch.randn(N, 1, generator=g) * 3.0
        dones = (torch.rand(N, generator=g) < 0.08).to(torch.uint8)
        if t == 3:
            dones.zero_()  # a step without finished episodes
        current_rewards += rewards
        current_lengths += 1
        all_done_indices = dones.nonzero(as_tuple=False)
        done_indices = all_done_indices[::1]
        vals_r = current_rewards[done_indices]
        vals_l = current_lengths[done_indices]
        out["n_done"].append(np.int64(vals_r.numel()))
        out["mean_reward"].append(vals_r.mean().numpy() if vals_r.numel() else np.float32("nan"))
        out["mean_length"].append(vals_l.mean().numpy() if vals_l.numel() else np.float32("nan"))
        out["sum_reward"].append(vals_r.double().sum().numpy())
        out["sum_length"].append(vals_l.double().sum().numpy())
        not_dones = 1.0 - dones.float()
        current_rewards = current_rewards * not_dones.unsqueeze(1)
        current_lengths = current_lengths * not_dones
        out["rew"].append(rewards.squeeze(1).numpy().copy())
        out["dones"].append(dones.numpy().astype(np.int64))
        out["current_rewards"].append(current_rewards.squeeze(1).numpy().copy())
        out["current_lengths"].append(current_lengths.numpy().copy())
    res = {k: np.stack([np.asarray(x) for x in v]) for k, v in out.items()}
    res["meta"] = _meta(task="EpisodeStats", N=N, T=T, seed=seed)
    np.savez_compressed(os.path.join(OUT, "episode_stats.npz"), **res)
    print("episode_stats.npz", {k: v.shape for k, v in res.items() if hasattr(v, "shape")}, "done", int(res["n_done"].sum()))


def gen_observer(seed=41, N=333, H=5, EPOCHS=4, G=48):
    """Episode statistics from the reference's own code: CommonAgent.play_steps (learning/common_agent.py:250-310)
    runs its loop - current_rewards / current_lengths bookkeeping, game_rewards.update(...),
    algo_observer.process_infos(infos, done_indices) - with the reference's RLGPUAlgoObserver
    (utils/rlgames_utils.py:132-211) attached, and after_print_stats once per epoch.  The agent is the
    reference class without its constructor; environment, policy, critic and experience buffer are
    placeholders; what game_rewards / game_lengths receive and what the observer writes is recorded."""
    from unittest import mock
    ref_shim.install_learning_stubs()
    ru = ref_shim.load("isaacgymenvs.utils.rlgames_utils")
    ca = ref_shim.load("isaacgymenvs.learning.common_agent")
    g = torch.Generator().manual_seed(seed)

    class Meter:  # records what rl_games' AverageMeter.update would be given
        def __init__(self):
            self.calls = []

        def update(self, values):
            self.calls.append(values.reshape(-1).clone())

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, float(value), float(step)))

    class Agent(ca.CommonAgent):
        def __init__(self):
            pass

        def set_eval(self):
            pass

        def _env_reset_done(self):
            return {"obs": torch.zeros(N, 1)}, torch.zeros(0, dtype=torch.long)

        def get_action_values(self, obs):
            return {"actions": torch.zeros(N, 1)}

        def _eval_critic(self, obs):
            return torch.zeros(N, 1)

        def discount_values(self, *a):
            return mock.MagicMock()

        def env_step(self, actions):
            return self._feed.pop(0)

    a = Agent()
    a.update_list, a.horizon_length, a.use_action_masks, a.has_central_value = [], H, False, False
    a.experience_buffer = mock.MagicMock()
    a.rewards_shaper = lambda r: r
    a.num_agents, a.batch_size, a.tensor_list = 1, N * H, []
    a.current_rewards = torch.zeros(N, 1)
    a.current_lengths = torch.zeros(N)
    a.game_rewards, a.game_lengths = Meter(), Meter()
    a.games_to_track, a.device = G, "cpu"
    a.writer = Writer()
    obs = ru.RLGPUAlgoObserver()
    obs.after_init(a)
    a.algo_observer = obs

    keys = ("successes", "energy")
    rec = {k: [] for k in ("rew", "dones", "cum_successes", "cum_energy", "ep_level", "ep_bonus", "cons", "current_rewards",
                           "current_lengths", "acc_successes", "acc_energy", "n_done", "sum_reward", "sum_length",
                           "mean_reward", "mean_length")}
    windows, writes = [], []
    for epoch in range(EPOCHS):
        feed = []
        for t in range(H):
            rewards = torch.randn(N, 1, generator=g) * 3.0
            p_done = 0.5 if (epoch == 1 and t == 2) else 0.07   # one step finishes more episodes than the window holds
            dones = (torch.rand(N, generator=g) < p_done).to(torch.uint8)
            if epoch == 2:
                dones.zero_()  # an epoch without a finished episode: nothing new is logged
            infos = {"terminate": torch.zeros(N), "consecutive_successes": torch.rand((), generator=g),
                     "episode": {"level": torch.rand((), generator=g), "bonus": torch.rand(3, generator=g)},
                     "episode_cumulative": {"successes": (torch.rand(N, generator=g) < 0.3).float(),
                                            "energy": torch.randn(N, generator=g)},
                     "scalars": {"lr": 0.25 * (epoch + 1)}, "tensor_info": torch.zeros(N)}
            feed.append(({"obs": torch.zeros(N, 1)}, rewards, dones, infos))
            rec["rew"].append(rewards.squeeze(1).numpy().copy())
            rec["dones"].append(dones.numpy().astype(np.int64))
            rec["cum_successes"].append(infos["episode_cumulative"]["successes"].numpy().copy())
            rec["cum_energy"].append(infos["episode_cumulative"]["energy"].numpy().copy())
            rec["ep_level"].append(infos["episode"]["level"].numpy().copy())
            rec["ep_bonus"].append(infos["episode"]["bonus"].numpy().copy())
            rec["cons"].append(infos["consecutive_successes"].numpy().copy())
        a._feed = feed
        n_before = len(a.game_rewards.calls)

        # per-step state: hook the observer (the reference calls it once per step, last thing but the masking)
        real = obs.process_infos

        def spy(infos, done_indices, _real=real):
            _real(infos, done_indices)
            rec["acc_successes"].append(obs.episode_cumulative["successes"].numpy().copy())
            rec["acc_energy"].append(obs.episode_cumulative["energy"].numpy().copy())
        obs.process_infos = spy
        a.play_steps()
        obs.process_infos = real
        for vr, vl in zip(a.game_rewards.calls[n_before:], a.game_lengths.calls[n_before:]):
            rec["n_done"].append(np.int64(vr.numel()))
            rec["sum_reward"].append(vr.double().sum().numpy())
            rec["sum_length"].append(vl.double().sum().numpy())
            rec["mean_reward"].append(vr.mean().numpy() if vr.numel() else np.float32("nan"))
            rec["mean_length"].append(vl.mean().numpy() if vl.numel() else np.float32("nan"))
        # the agent masks after the observer ran: state after the whole step is only visible per epoch
        rec["current_rewards"].append(a.current_rewards.squeeze(1).numpy().copy())
        rec["current_lengths"].append(a.current_lengths.numpy().copy())
        windows.append({k: [float(x) for x in obs.episode_cumulative_avg.get(k, [])] for k in keys})
        a.writer.rows = []
        obs.after_print_stats(frame=(epoch + 1) * N * H, epoch_num=epoch + 1, total_time=1.5 * (epoch + 1))
        writes.append(list(a.writer.rows))
    res = {k: np.stack([np.asarray(x) for x in v]) for k, v in rec.items()}
    res["meta"] = _meta(task="Observer", N=N, H=H, EPOCHS=EPOCHS, G=G, seed=seed, keys=list(keys), windows=windows,
                        writes=writes)
    np.savez_compressed(os.path.join(OUT, "observer.npz"), **res)
    print("observer.npz", {k: v.shape for k, v in res.items() if hasattr(v, "shape")}, "done", int(res["n_done"].sum()),
          "rows", [len(w) for w in writes])


def gen_angles_edge(seed=81, n=2560):
    """Torso rotations where the numerators of the roll / yaw atan2 of get_euler_xyz, 2(wx+yz) and 2(wz+xy), are
    zero, signed zero, subnormal or at rounding-noise level, of both signs: `remainder(atan2(.), 2 pi)` maps a
    numerator of -1e-9 to 6.2831855 and +1e-9 to 1e-9, so a sign that differs from the reference's is an error
    of 2 pi in an observation.  Recorded: the reference's get_euler_xyz, and the twelve leading observation
    columns of compute_ant_observations / compute_humanoid_observations (identity and tilted start rotation)."""
    _install_gym_stubs()
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    ant = ref_shim.load("isaacgymenvs.tasks.ant")
    hum = ref_shim.load("isaacgymenvs.tasks.humanoid")
    rs = np.random.default_rng(seed)
    deltas = np.array([0.0, 1e-9, -1e-9, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 3e-10, -3e-10])
    tiny = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -1e-40, 1e-38, -1e-38, 1e-30, -1e-30, 1e-20, -1e-20, 1e-12, -1e-12,
                     1e-9, -1e-9, 1e-7, -1e-7])
    m = n // 5

    def solve(kind, count):
        """unit quaternions (fp64) with w x + y z = delta (kind 0) or w z + x y = delta (kind 1)"""
        a, b = rs.normal(size=count), rs.normal(size=count)
        w = rs.uniform(0.2, 1.0, size=count) * rs.choice([-1.0, 1.0], size=count)
        d = rs.choice(deltas, size=count)
        c = (d - a * b) / w
        q = np.stack([c, a, b, w], -1) if kind == 0 else np.stack([a, b, c, w], -1)
        return q / np.linalg.norm(q, axis=-1, keepdims=True)

    qa = solve(0, m)                       # roll numerator at noise level
    qb = solve(1, m)                       # yaw numerator at noise level
    # pure pitch plus tiny x and z: both numerators are 2 w delta, exactly representable tiny values
    pitch = rs.uniform(-np.pi, np.pi, size=m)
    qc = np.stack([rs.choice(tiny, size=m), np.sin(pitch / 2), rs.choice(tiny, size=m), np.cos(pitch / 2)], -1)
    # roll / yaw near pi: the denominator is negative, the numerator tiny (atan2 -> +-pi)
    ang = np.pi + rs.choice(deltas, size=m) * 10
    half = rs.random(m) < 0.5
    qd = np.where(half[:, None], np.stack([np.sin(ang / 2), 0 * ang, 0 * ang, np.cos(ang / 2)], -1),
                  np.stack([0 * ang, 0 * ang, np.sin(ang / 2), np.cos(ang / 2)], -1))
    qe = rs.normal(size=(n - 4 * m, 4))    # generic rotations, not normalised to the last bit
    qe /= np.linalg.norm(qe, axis=-1, keepdims=True)
    q = np.concatenate([qa, qb, qc, qd, qe]).astype(np.float32)
    special = np.array([[0, 0, 0, 1], [0, 0, 0, -1], [0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [-0.0, -0.0, 0, 1],
                        [-0.0, 0, 0, -1], [0, -0.0, -0.0, 1], [0, 0, -0.0, -1], [-0.0, -0.0, -0.0, -1],
                        [0.70710678, 0, 0, 0.70710678], [0, 0.70710678, 0, 0.70710678], [0, 0, 0.70710678, -0.70710678],
                        [0.5, 0.5, 0.5, 0.5], [0.5, -0.5, 0.5, -0.5]], np.float32)
    q[-len(special):] = special
    q = torch.from_numpy(q)
    root = torch.zeros(n, 13)
    root[:, 0:3] = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32)) * torch.tensor([3.0, 3.0, 0.2]) + torch.tensor([0.0, 0.0, 1.0])
    root[:, 3:7] = q
    root[:, 7:13] = torch.from_numpy(rs.normal(size=(n, 6)).astype(np.float32))
    out = {"root": root}
    r, p_, y = tj.get_euler_xyz(q)
    out.update(euler_roll=r, euler_pitch=p_, euler_yaw=y)
    targets1 = torch.tensor([1000.0, 0.0, 0.0])
    tilt = torch.tensor([0.1, -0.2, 0.3, 0.9])
    tilt = tilt / tilt.norm()
    basis0, basis1 = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1), torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1)
    for tag, start_rot in (("", torch.tensor([0.0, 0.0, 0.0, 1.0])), ("_tilt", tilt)):
        isr = tj.quat_conjugate(start_rot).repeat(n, 1)
        for name, D, S in (("ant", 8, 24), ("hum", 21, 12)):
            lo, hi = -torch.ones(D), torch.ones(D)
            zeros = torch.zeros(n, D)
            pot = torch.zeros(n)
            if name == "ant":
                obs = ant.compute_ant_observations(torch.zeros(n, 60), root.clone(), targets1.repeat(n, 1), pot, isr, zeros, zeros, lo, hi,
                                                   0.2, torch.zeros(n, S), zeros, 0.0166, 0.1, basis0, basis1, 2)[0]
            else:
                obs = hum.compute_humanoid_observations(torch.zeros(n, 108), root.clone(), targets1.repeat(n, 1), pot, isr, zeros, zeros,
                                                        zeros, lo, hi, 0.1, torch.zeros(n, S), zeros, 0.0166, 0.01, 0.25, basis0,
                                                        basis1)[0]
            out["%s_obs_head%s" % (name, tag)] = obs[:, :12].clone()
        out["inv_start_rot" + tag] = isr[0].clone()
    out = {k: v.numpy() for k, v in out.items()}
    sr = 2.0 * (out["root"][:, 6] * out["root"][:, 3] + out["root"][:, 4] * out["root"][:, 5])
    sy = 2.0 * (out["root"][:, 6] * out["root"][:, 5] + out["root"][:, 3] * out["root"][:, 4])
    out["meta"] = _meta(task="angles_edge", n=n, seed=seed, dt=0.0166, targets=[1000.0, 0.0, 0.0])
    np.savez_compressed(os.path.join(OUT, "angles_edge.npz"), **out)
    print("angles_edge.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")},
          "|roll numerator| < 1e-6: %d (%d negative), |yaw numerator| < 1e-6: %d (%d negative); rolls at 2 pi: %d, yaws at 2 pi: %d" % (
              (np.abs(sr) < 1e-6).sum(), ((np.abs(sr) < 1e-6) & (sr < 0)).sum(), (np.abs(sy) < 1e-6).sum(),
              ((np.abs(sy) < 1e-6) & (sy < 0)).sum(), (out["euler_roll"] > 6.28).sum(), (out["euler_yaw"] > 6.28).sum()))


def gen_amp_demo(seed=11):
    """HumanoidAMP.fetch_amp_obs_demo of the reference class (tasks/humanoid_amp.py:146-171) on the motion set of
    motion.npz: numpy-seeded draws (replayed and stored), the tiling over the history steps, get_motion_state and
    build_amp_observations, for two / three history steps and both root-observation frames."""
    _install_gym_stubs()
    import tempfile
    sk = ref_shim.load("isaacgymenvs.tasks.amp.poselib.poselib.skeleton.skeleton3d")
    ml = ref_shim.load("isaacgymenvs.tasks.amp.utils_amp.motion_lib")
    hamp = ref_shim.load("isaacgymenvs.tasks.humanoid_amp")
    tree = sk.SkeletonTree(AMP_NODE_NAMES, torch.tensor(AMP_PARENTS), torch.tensor(AMP_LOCAL_T, dtype=torch.float32))
    specs = [(90, 30, 1.0), (61, 60, 3.0)]  # the motion set of gen_motion
    with tempfile.TemporaryDirectory() as d:
        entries = []
        for i, (F, fps, wgt) in enumerate(specs):
            _synth_motion(sk, tree, F, fps, seed + i).to_file(os.path.join(d, "m%d.npy" % i))
            entries.append({"file": "m%d.npy" % i, "weight": wgt})
        with open(os.path.join(d, "set.yaml"), "w") as f:
            yaml.safe_dump({"motions": entries}, f)
        cwd = os.getcwd()
        os.chdir("/")
        lib = ml.MotionLib(motion_file=os.path.join(d, "set.yaml"), num_dofs=28, key_body_ids=np.array([5, 8, 11, 14]),
                           device="cpu")
        os.chdir(cwd)
    have = np.load(os.path.join(OUT, "motion.npz"), allow_pickle=False)
    assert np.array_equal(have["motion_lengths"], lib._motion_lengths)
    assert np.array_equal(have["m1_local_rotation"], lib._motions[1].local_rotation.numpy())

    class Fake(hamp.HumanoidAMP):
        def __init__(self):
            pass

    out, cases = {}, []
    for c, (K, lro, dt, n, np_seed) in enumerate([(2, False, 0.0332, 160, 31), (3, True, 0.0166, 96, 32), (4, False, 0.05, 64, 33)]):
        t = Fake()
        t.device, t._motion_lib, t.dt, t._num_amp_obs_steps, t._local_root_obs = "cpu", lib, dt, K, lro
        t._amp_obs_demo_buf, t.num_amp_obs = None, K * 105
        np.random.seed(np_seed)
        demo = t.fetch_amp_obs_demo(n)
        np.random.seed(np_seed)
        ids = lib.sample_motions(n)
        times0 = lib.sample_time(ids)
        out["c%d_ids" % c], out["c%d_times0" % c], out["c%d_demo" % c] = ids.copy(), times0.copy(), demo.numpy().copy()
        cases.append(dict(K=K, local_root_obs=lro, dt=dt, n=n, np_seed=np_seed))
        assert demo.shape == (n, K * 105)
    out["meta"] = _meta(task="HumanoidAMP.fetch_amp_obs_demo", seed=seed, cases=cases)
    np.savez_compressed(os.path.join(OUT, "amp_demo.npz"), **out)
    print("amp_demo.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def gen_math_nonfinite():
    """Every torch_jit_utils function on inputs with NaN / +-Inf rows (rows 100 ... of each input)."""
    gen_math(seed=13, n=192, name="math_nonfinite", P=_poison_rows)
    gen_math2(seed=14, n=192, name="math2_nonfinite", P=_poison_rows)


def gen_nonfinite_tasks():
    """Cartpole and AllegroHand on non-finite simulator state / actions (HumanoidAMP's comes out of gen_motion)."""
    gen_cartpole(seed=93, N=128, T=4, name="cartpole_nonfinite", poison=True)
    _gen_allegro("nonfinite", N=96, T=4, seed=94, poison=True)
    _gen_shadow("nonfinite", N=70, T=3, seed=95, poison=True)


def gen_nonunit():
    """Ant, Humanoid, AllegroHand and ShadowHand with root / object quaternions of length 0.5 ... 2
    (HumanoidAMP's comes out of gen_motion)."""
    _gen_loco("ant_nonunit", "isaacgymenvs.tasks.ant", "Ant", "Ant", 8, 24, False, 101, 160, 3, poison="nonunit")
    _gen_loco("humanoid_nonunit", "isaacgymenvs.tasks.humanoid", "Humanoid", "Humanoid", 21, 12, True, 102, 128, 3, poison="nonunit")
    _gen_allegro("nonunit", N=96, T=3, seed=103, poison="nonunit")
    _gen_shadow("nonunit", N=70, T=3, seed=104, poison="nonunit")


def gen_imitate_extreme():
    """AntImitate and HumanoidImitate on extreme values (run with PYTORCH_JIT=0, as humanoid_imitate needs)."""
    gen_ant_imitate(seed=43, N=192, T=4, F=9, name="ant_imitate_extreme", poison="extreme")
    gen_humanoid_imitate(seed=53, N=80, T=3, F=48, name="humanoid_imitate_extreme", poison="extreme")


def gen_extreme_tasks():
    """Cartpole, AllegroHand, ShadowHand and every torch_jit_utils function on huge / tiny / subnormal / signed-zero /
    non-finite values scattered over the simulator state, the actions and the functions' inputs (Ant / Humanoid:
    gen_loco_extreme; HumanoidAMP's comes out of gen_motion)."""
    gen_cartpole(seed=96, N=256, T=3, name="cartpole_extreme", poison="extreme")
    _gen_allegro("extreme", N=128, T=3, seed=97, poison="extreme")
    _gen_shadow("extreme", N=96, T=3, seed=98, poison="extreme")
    gx = torch.Generator().manual_seed(99)

    def scatter(x):
        if x.is_floating_point() and x.numel() > 8:
            _scatter_extremes(gx, [x], 0.03)
        return x

    gen_math(seed=15, n=384, name="math_extreme", P=scatter)
    gen_math2(seed=16, n=384, name="math2_extreme", P=scatter)


def gen_extract_positions(seed=77):
    """extract_positions_from_rigid_body_states (utils/torch_jit_utils.py:670-681) on [n, 286] and wider rows.
    Run with PYTORCH_JIT=0, like humanoid_imitate: the function builds its index tensor on "cuda:0", and un-scripted
    that one allocation can be redirected to the CPU while the rest runs as written."""
    assert os.environ.get("PYTORCH_JIT") == "0", "run as: PYTORCH_JIT=0 python oracle/gen_golden.py extract_positions"
    _install_gym_stubs()
    tj = ref_shim.load("isaacgymenvs.utils.torch_jit_utils")
    real_empty = torch.empty

    def cpu_empty(*a, **k):
        if str(k.get("device", "")) == "cuda:0":
            k["device"] = "cpu"
        return real_empty(*a, **k)

    g = torch.Generator().manual_seed(seed)
    out = {}
    torch.empty = cpu_empty
    try:
        for name, n, w in (("a", 67, 286), ("b", 1, 286), ("c", 33, 300)):
            x = torch.randn(n, w, generator=g)
            x[0, 0], x[-1, -1] = float("nan"), float("inf")
            out["in_" + name] = x.numpy().copy()
            out["out_" + name] = tj.extract_positions_from_rigid_body_states(x).numpy().copy()
    finally:
        torch.empty = real_empty
    out["meta"] = np.array(_meta(note="reference function run with PYTORCH_JIT=0, its cuda:0 index tensor redirected to the CPU"))
    np.savez_compressed(os.path.join(OUT, "extract_positions.npz"), **out)
    print("extract_positions.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


GENERATORS = {"nonunit": gen_nonunit, "imitate_extreme": gen_imitate_extreme, "extreme_tasks": gen_extreme_tasks, "loco_extreme": gen_loco_extreme, "extract_positions": gen_extract_positions, "nonfinite_tasks": gen_nonfinite_tasks, "math_nonfinite": gen_math_nonfinite, "loco_nonfinite": gen_loco_nonfinite, "amp_demo": gen_amp_demo, "angles_edge": gen_angles_edge, "observer": gen_observer, "hand_forces": gen_hand_forces, "vectask": gen_vectask, "loco_alt": gen_loco_alt, "shadow": gen_shadow, "prestep": gen_prestep, "humanoid_imitate": gen_humanoid_imitate, "ant_imitate": gen_ant_imitate, "episode_stats": gen_episode_stats, "cartpole_sample": gen_cartpole_sample, "motion": gen_motion, "allegro": gen_allegro, "cartpole": gen_cartpole, "ant": gen_ant, "humanoid": gen_humanoid, "math": gen_math, "math2": gen_math2}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or [n for n in GENERATORS if n not in ("humanoid_imitate", "extract_positions", "imitate_extreme")]  # those need PYTORCH_JIT=0
    names.sort(key=lambda n: n != "motion")  # amp_demo reads the clips motion.npz holds: that file first
    for n in names:
        GENERATORS[n]()
