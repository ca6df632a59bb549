"""Seeded synthetic state tensors standing in for the physics engine.

The closed-source simulator is out of scope; parity tests and the benchmark
feed the step kernels with tensors drawn here (distributions chosen so that
every branch is exercised: ~0.5 % of envs below the termination height, a few
% of joints beyond the soft limits, ~0.1 % of envs timing out per step).
"""
import torch

from . import _abi

ANT = dict(num_dof=8, num_sensor_floats=24, num_obs=60, z_mean=0.55, z_std=0.1, humanoid=False,
           dt=0.0166, dof_vel_scale=0.2, contact_force_scale=0.1, angular_velocity_scale=0.1,
           up_weight=0.1, heading_weight=0.5, actions_cost_scale=0.005, energy_cost_scale=0.05,
           joints_at_limit_cost_scale=0.1, termination_height=0.31, death_cost=-2.0,
           max_episode_length=1000)
HUMANOID = dict(num_dof=21, num_sensor_floats=12, num_obs=108, z_mean=1.34, z_std=0.3, humanoid=True,
                dt=0.0166, dof_vel_scale=0.1, contact_force_scale=0.01, angular_velocity_scale=0.25,
                up_weight=0.1, heading_weight=0.5, actions_cost_scale=0.01, energy_cost_scale=0.05,
                joints_at_limit_cost_scale=0.25, termination_height=0.8, death_cost=-1.0,
                max_episode_length=1000)


def loco_limits(D):
    lo = [-1.0 - 0.01 * j for j in range(D)]
    hi = [1.0 + 0.02 * j for j in range(D)]
    return lo, hi


def loco_params(spec, seed=0, rng_offset=0):
    D = spec["num_dof"]
    lo, hi = loco_limits(D)
    p = _abi.LocoParams(
        num_dof=D, num_sensor_floats=spec["num_sensor_floats"], dt=spec["dt"],
        dof_vel_scale=spec["dof_vel_scale"], contact_force_scale=spec["contact_force_scale"],
        angular_velocity_scale=spec["angular_velocity_scale"], up_weight=spec["up_weight"],
        heading_weight=spec["heading_weight"], actions_cost_scale=spec["actions_cost_scale"],
        energy_cost_scale=spec["energy_cost_scale"],
        joints_at_limit_cost_scale=spec["joints_at_limit_cost_scale"],
        termination_height=spec["termination_height"], death_cost=spec["death_cost"],
        max_episode_length_m1=float(spec["max_episode_length"] - 1),
        targets=[1000.0, 0.0, 0.0], inv_start_rot=[-0.0, -0.0, -0.0, 1.0],
        basis_vec0=[1.0, 0.0, 0.0], basis_vec1=[0.0, 0.0, 1.0],
        dof_limits_lower=lo, dof_limits_upper=hi, seed=seed, rng_offset=rng_offset)
    if spec["humanoid"]:
        efforts = [20.0 + 180.0 * ((7 * j) % D) / D for j in range(D)]
        p.set("motor_efforts", efforts)
        p.max_motor_effort = max(efforts)
    return p


def _quats(n, gen, device):
    q = torch.randn(n, 4, generator=gen, device=device)
    small = torch.randn(n, 4, generator=gen, device=device) * 1e-3
    small[:, 3] = 1.0
    pick = torch.rand(n, generator=gen, device=device) < 0.5
    q = torch.where(pick.unsqueeze(-1), q, small)
    return q / q.norm(dim=-1, keepdim=True)


def loco_state(spec, n, seed, device, with_rand=False):
    """Returns the name->tensor dict of igx_loco_buffers for n environments."""
    D, S = spec["num_dof"], spec["num_sensor_floats"]
    gen = torch.Generator(device=device).manual_seed(seed)
    lo, hi = loco_limits(D)
    lo = torch.tensor(lo, device=device)
    hi = torch.tensor(hi, device=device)
    root = torch.empty(n, 13, device=device)
    root[:, 0:3] = torch.randn(n, 3, generator=gen, device=device) * 0.1
    root[:, 2] = spec["z_mean"] + torch.randn(n, generator=gen, device=device) * spec["z_std"]
    root[:, 3:7] = _quats(n, gen, device)
    root[:, 7:13] = torch.randn(n, 6, generator=gen, device=device)
    dof_state = torch.empty(n, D, 2, device=device)
    u = torch.rand(n, D, generator=gen, device=device) * 1.04 - 0.02
    dof_state[..., 0] = lo + (hi - lo) * u
    dof_state[..., 1] = torch.randn(n, D, generator=gen, device=device)
    init_root = torch.zeros(n, 13, device=device)
    init_root[:, 0:2] = torch.randn(n, 2, generator=gen, device=device) * 0.05
    init_root[:, 2] = spec["z_mean"]
    init_root[:, 6] = 1.0
    t = dict(
        root_states=root, dof_state=dof_state,
        sensors=torch.randn(n, S, generator=gen, device=device) * 10.0,
        actions=torch.rand(n, D, generator=gen, device=device) * 2.0 - 1.0,
        initial_root_states=init_root,
        initial_dof_pos=torch.zeros(n, D, device=device),
        obs=torch.zeros(n, spec["num_obs"], device=device),
        potentials=torch.full((n,), -1000.0 / spec["dt"], device=device),
        prev_potentials=torch.full((n,), -1000.0 / spec["dt"], device=device),
        up_vec=torch.zeros(n, 3, device=device), heading_vec=torch.zeros(n, 3, device=device),
        rew=torch.zeros(n, device=device),
        reset=(torch.rand(n, generator=gen, device=device) < 0.01).to(torch.int64),
        progress=torch.randint(0, spec["max_episode_length"], (n,), generator=gen, device=device),
        randomize=torch.zeros(n, dtype=torch.int64, device=device),
    )
    if spec["humanoid"]:
        t["dof_force"] = torch.randn(n, D, generator=gen, device=device) * 30.0
    if with_rand:
        t["rand"] = torch.rand(n, 2 * D, generator=gen, device=device)
    return t


def cartpole_state(n, seed, device, with_rand=False):
    gen = torch.Generator(device=device).manual_seed(seed)
    ds = torch.empty(n, 2, 2, device=device)
    ds[:, 0, 0] = (torch.rand(n, generator=gen, device=device) * 2 - 1) * 3.5
    ds[:, 1, 0] = (torch.rand(n, generator=gen, device=device) * 2 - 1) * 2.0
    ds[:, :, 1] = torch.randn(n, 2, generator=gen, device=device)
    t = dict(dof_state=ds, obs=torch.zeros(n, 4, device=device), rew=torch.zeros(n, device=device),
             reset=(torch.rand(n, generator=gen, device=device) < 0.05).to(torch.int64),
             progress=torch.randint(0, 500, (n,), generator=gen, device=device))
    if with_rand:
        t["rand"] = torch.rand(n, 4, generator=gen, device=device)
    return t


def cartpole_params(seed=0, rng_offset=0):
    return _abi.CartpoleParams(reset_dist=3.0, max_episode_length_m1=499.0, seed=seed, rng_offset=rng_offset)


# ---------------------------------------------------------------- AllegroHand
ALLEGRO_SUCCESS_C = {0.1: 0.04997916892170906, 0.2: 0.0998334139585495}


def allegro_params(obs_type="full_state", seed=0, rng_offset=0, **over):
    D = _abi.IGX_ALLEGRO_DOF
    lo = [-0.3 - 0.02 * j for j in range(D)]
    hi = [0.6 + 0.05 * j for j in range(D)]
    kw = dict(obs_type=_abi.ALLEGRO_OBS_TYPES[obs_type], max_consecutive_successes=0, ignore_z_rot=0,
              use_relative_control=0, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1,
              action_penalty_scale=-0.0002, success_tolerance=0.1, success_threshold_c=ALLEGRO_SUCCESS_C[0.1],
              reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0, max_episode_length_m1=599.0,
              av_factor=0.1, one_minus_av_factor=1.0 - float(torch.tensor(0.1, dtype=torch.float32)),
              vel_obs_scale=0.2, force_torque_obs_scale=10.0, reset_position_noise=0.01,
              reset_dof_pos_noise=0.2, reset_dof_vel_noise=0.0, act_moving_average=1.0,
              one_minus_act_moving_average=0.0, dof_speed_scale_dt=20.0 * 0.01667,
              goal_displacement=[-0.2, -0.06, 0.12], dof_lower=lo, dof_upper=hi, seed=seed, rng_offset=rng_offset)
    kw.update(over)
    return _abi.AllegroParams(**kw)


def allegro_workspace_floats(n):
    """IGX_ALLEGRO_WORKSPACE_BYTES(n) / 4."""
    return 4 * ((n + 15) // 16 + 1)


def allegro_state(n, seed, device, obs_type="full_state", asymmetric=False):
    """name->tensor dict of igx_allegro_buffers; ~1 % of envs start within the
    success tolerance, object/goal distance up to 0.3 m (fallDistance 0.24)."""
    D = _abi.IGX_ALLEGRO_DOF
    gen = torch.Generator(device=device).manual_seed(seed)

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True)

    lo = torch.tensor([-0.3 - 0.02 * j for j in range(D)], device=device)
    hi = torch.tensor([0.6 + 0.05 * j for j in range(D)], device=device)
    dof_state = torch.empty(n, D, 2, device=device)
    dof_state[..., 0] = lo + (hi - lo) * (torch.rand(n, D, generator=gen, device=device) * 1.04 - 0.02)
    dof_state[..., 1] = torch.randn(n, D, generator=gen, device=device)
    obj_init = torch.zeros(n, 13, device=device)
    obj_init[:, 0:3] = torch.tensor([0.0, -0.2, 0.56], device=device) + 0.01 * torch.randn(n, 3, generator=gen, device=device)
    obj_init[:, 6] = 1.0
    goal_init = obj_init.clone()
    goal_init[:, 2] -= 0.04
    goal_states = goal_init.clone()
    goal_states[:, 3:7] = unit(torch.randn(n, 4, generator=gen, device=device))
    root = torch.zeros(3 * n, 13, device=device)
    off = torch.randn(n, 3, generator=gen, device=device)
    off = off / off.norm(dim=-1, keepdim=True) * (torch.rand(n, 1, generator=gen, device=device) * 0.3)
    obj = torch.zeros(n, 13, device=device)
    obj[:, 0:3] = goal_states[:, 0:3] + off
    q = unit(torch.randn(n, 4, generator=gen, device=device))
    near = torch.rand(n, generator=gen, device=device) < 0.01
    q = torch.where(near.unsqueeze(-1), unit(goal_states[:, 3:7] + 0.02 * torch.randn(n, 4, generator=gen, device=device)), q)
    obj[:, 3:7] = q
    obj[:, 7:13] = torch.randn(n, 6, generator=gen, device=device)
    idx = torch.arange(n, device=device, dtype=torch.int64)
    root[3 * idx + 1] = obj
    nobs = _abi.ALLEGRO_NUM_OBS[_abi.ALLEGRO_OBS_TYPES[obs_type]]
    return dict(
        dof_state=dof_state, dof_force=torch.randn(n, D, generator=gen, device=device),
        root_state_tensor=root, object_indices=3 * idx + 1, goal_object_indices=3 * idx + 2,
        goal_states=goal_states, goal_init_state=goal_init, object_init_state=obj_init,
        actions=torch.rand(n, D, generator=gen, device=device) * 2 - 1,
        prev_targets=torch.zeros(n, D, device=device), cur_targets=torch.zeros(n, D, device=device),
        obs=torch.zeros(n, nobs, device=device),
        states=torch.zeros(n, 88, device=device) if asymmetric else None,
        rew=torch.zeros(n, device=device),
        reset=(torch.rand(n, generator=gen, device=device) < 0.01).to(torch.int64),
        reset_goal=(torch.rand(n, generator=gen, device=device) < 0.01).to(torch.int64),
        progress=torch.randint(0, 600, (n,), generator=gen, device=device),
        randomize=torch.zeros(n, dtype=torch.int64, device=device),
        successes=torch.randint(0, 5, (n,), generator=gen, device=device).float(),
        consecutive_successes=torch.zeros(1, device=device),
        workspace=torch.empty(allegro_workspace_floats(n), device=device))


# ---------------------------------------------------------------- ShadowHand
SHADOW_ACTUATED = [j for j in range(24) if j not in (5, 9, 13, 18)]   # the coupled distal finger joints are passive
SHADOW_FINGERTIPS = [7, 11, 15, 20, 25]
SHADOW_NUM_BODIES = 28


def shadow_params(obs_type="full_state", seed=0, rng_offset=0, **over):
    D = _abi.IGX_SHADOW_DOF
    lo = [-0.3 - 0.02 * j for j in range(D)]
    hi = [0.6 + 0.05 * j for j in range(D)]
    kw = dict(obs_type=_abi.IGX_SHADOW_OBS[obs_type], max_consecutive_successes=0, ignore_z_rot=0,
              use_relative_control=0, dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1,
              action_penalty_scale=-0.0002, success_tolerance=0.1, success_threshold_c=ALLEGRO_SUCCESS_C[0.1],
              reach_goal_bonus=250.0, fall_dist=0.24, fall_penalty=0.0, max_episode_length_m1=599.0,
              av_factor=0.1, one_minus_av_factor=1.0 - float(torch.tensor(0.1, dtype=torch.float32)),
              vel_obs_scale=0.2, force_torque_obs_scale=10.0, reset_position_noise=0.01,
              reset_dof_pos_noise=0.2, reset_dof_vel_noise=0.0, act_moving_average=1.0,
              one_minus_act_moving_average=0.0, dof_speed_scale_dt=20.0 * 0.01667,
              goal_displacement=[-0.2, -0.06, 0.12], num_bodies=SHADOW_NUM_BODIES, dof_lower=lo, dof_upper=hi,
              actuated_dof_indices=SHADOW_ACTUATED, fingertip_handles=SHADOW_FINGERTIPS, seed=seed, rng_offset=rng_offset)
    kw.update(over)
    return _abi.ShadowParams(**kw)


def shadow_state(n, seed, device, obs_type="full_state", asymmetric=False):
    """name->tensor dict of igx_shadow_buffers, drawn like allegro_state."""
    D, A, NB = _abi.IGX_SHADOW_DOF, _abi.IGX_SHADOW_ACTIONS, SHADOW_NUM_BODIES
    gen = torch.Generator(device=device).manual_seed(seed)

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True)

    lo = torch.tensor([-0.3 - 0.02 * j for j in range(D)], device=device)
    hi = torch.tensor([0.6 + 0.05 * j for j in range(D)], device=device)
    dof_state = torch.empty(n, D, 2, device=device)
    dof_state[..., 0] = lo + (hi - lo) * (torch.rand(n, D, generator=gen, device=device) * 1.04 - 0.02)
    dof_state[..., 1] = torch.randn(n, D, generator=gen, device=device)
    obj_init = torch.zeros(n, 13, device=device)
    obj_init[:, 0:3] = torch.tensor([0.0, -0.39, 0.6], device=device) + 0.01 * torch.randn(n, 3, generator=gen, device=device)
    obj_init[:, 6] = 1.0
    goal_init = obj_init.clone()
    goal_init[:, 2] -= 0.04
    goal_states = goal_init.clone()
    goal_states[:, 3:7] = unit(torch.randn(n, 4, generator=gen, device=device))
    root = torch.zeros(3 * n, 13, device=device)
    off = torch.randn(n, 3, generator=gen, device=device)
    off = off / off.norm(dim=-1, keepdim=True) * (torch.rand(n, 1, generator=gen, device=device) * 0.3)
    obj = torch.zeros(n, 13, device=device)
    obj[:, 0:3] = goal_states[:, 0:3] + off
    q = unit(torch.randn(n, 4, generator=gen, device=device))
    near = torch.rand(n, generator=gen, device=device) < 0.01
    q = torch.where(near.unsqueeze(-1), unit(goal_states[:, 3:7] + 0.02 * torch.randn(n, 4, generator=gen, device=device)), q)
    obj[:, 3:7] = q
    obj[:, 7:13] = torch.randn(n, 6, generator=gen, device=device)
    idx = torch.arange(n, device=device, dtype=torch.int64)
    root[3 * idx + 1] = obj
    full = obs_type == "full_state" or asymmetric
    return dict(
        dof_state=dof_state, dof_force=torch.randn(n, D, generator=gen, device=device) if full else None,
        vec_sensor_tensor=torch.randn(n, 30, generator=gen, device=device) if full else None,
        rigid_body_states=torch.randn(n, NB, 13, generator=gen, device=device),
        root_state_tensor=root, object_indices=3 * idx + 1, goal_object_indices=3 * idx + 2,
        goal_states=goal_states, goal_init_state=goal_init, object_init_state=obj_init,
        actions=torch.rand(n, A, generator=gen, device=device) * 2 - 1,
        prev_targets=torch.zeros(n, D, device=device), cur_targets=torch.zeros(n, D, device=device),
        obs=torch.zeros(n, _abi.IGX_SHADOW_NUM_OBS[obs_type], device=device),
        states=torch.zeros(n, 211, device=device) if asymmetric else None,
        rew=torch.zeros(n, device=device),
        reset=(torch.rand(n, generator=gen, device=device) < 0.01).to(torch.int64),
        reset_goal=(torch.rand(n, generator=gen, device=device) < 0.01).to(torch.int64),
        progress=torch.randint(0, 600, (n,), generator=gen, device=device),
        randomize=torch.zeros(n, dtype=torch.int64, device=device),
        successes=torch.randint(0, 5, (n,), generator=gen, device=device).float(),
        consecutive_successes=torch.zeros(1, device=device),
        workspace=torch.empty(allegro_workspace_floats(n), device=device))


# ---------------------------------------------------------------- HumanoidAMP
def amp_params(num_amp_obs_steps=2, local_root_obs=False, enable_early_termination=True, max_episode_length=300):
    return _abi.AmpParams(num_amp_obs_steps=num_amp_obs_steps, local_root_obs=int(local_root_obs),
                          enable_early_termination=int(enable_early_termination), num_contact_bodies=2,
                          contact_body_ids=[11, 14, 0, 0], key_body_ids=[5, 8, 11, 14], termination_height=0.5,
                          max_episode_length_m1=float(max_episode_length - 1))


def amp_state(n, seed, device, num_amp_obs_steps=2, max_episode_length=300):
    """name->tensor dict of igx_amp_buffers; ~3 % of envs carry a contact force on a
    non-foot body, about half of those also have a body below the termination height."""
    NB, ND, NO = _abi.IGX_AMP_NUM_BODIES, _abi.IGX_AMP_NUM_DOF, _abi.IGX_AMP_OBS
    gen = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=gen, device=device)  # noqa: E731
    u = lambda *s: torch.rand(*s, generator=gen, device=device)  # noqa: E731
    root = torch.empty(n, 13, device=device)
    root[:, 0:3] = r(n, 3) * 0.5
    root[:, 2] = 0.9 + r(n) * 0.1
    root[:, 3:7] = _quats(n, gen, device)
    root[:, 7:13] = r(n, 6)
    dof_state = torch.empty(n, ND, 2, device=device)
    dof_state[..., 0] = (u(n, ND) * 2 - 1) * 2.5
    dof_state[..., 0] = torch.where(u(n, ND) < 0.02, torch.zeros((), device=device), dof_state[..., 0])
    dof_state[..., 1] = r(n, ND) * 3.0
    body = torch.empty(n, NB, 13, device=device)
    body[..., 0:3] = root[:, None, 0:3] + r(n, NB, 3) * 0.3
    body[..., 2] = 0.9 + r(n, NB) * 0.2
    low = (u(n, NB) < 0.002)
    body[..., 2] = torch.where(low, 0.3 + 0.1 * r(n, NB), body[..., 2])
    body[..., 3:7] = _quats(n * NB, gen, device).view(n, NB, 4)
    body[..., 7:13] = r(n, NB, 6)
    cf = torch.zeros(n, NB, 3, device=device)
    hit = (u(n, NB) < 0.004) | low & (u(n, NB) < 0.5)
    cf = torch.where(hit.unsqueeze(-1), r(n, NB, 3) * 5.0, cf)
    cf[:, [11, 14]] = r(n, 2, 3) * 50.0  # feet always in contact
    return dict(
        root_states=root, dof_state=dof_state, rigid_body_state=body, contact_forces=cf,
        obs=torch.zeros(n, NO, device=device), amp_obs=r(n, num_amp_obs_steps, NO),
        rew=torch.zeros(n, device=device),
        reset=torch.zeros(n, dtype=torch.int64, device=device),
        terminate=torch.zeros(n, dtype=torch.int64, device=device),
        progress=torch.randint(0, max_episode_length + 1, (n,), generator=gen, device=device))


# ---------------------------------------------------------------- HumanoidImitate
def him_limits():
    D = _abi.IGX_HIM_DOF
    return [-1.0 - 0.01 * (j % 21) for j in range(D)], [1.0 + 0.02 * (j % 21) for j in range(D)]


def him_params(seed=0, rng_offset=0, termination_radius=1.0):
    lo, hi = him_limits()
    return _abi.HumanoidImitateParams(
        dt=0.0166, dof_vel_scale=0.1, up_weight=0.1, actions_cost_scale=0.01, termination_height=0.5,
        termination_radius_sq=float(torch.tensor(termination_radius ** 2, dtype=torch.float32)),
        max_episode_length_m1=999.0, targets=[1000.0, 0.0, 0.0], dof_limits_lower=lo, dof_limits_upper=hi,
        seed=seed, rng_offset=rng_offset)


def him_state(n, seed, device, with_rand=False):
    """name->tensor dict of igx_humanoid_imitate_buffers: ~12 % of roots outside the unit circle,
    a few % below the termination height."""
    D, NB, NO = _abi.IGX_HIM_DOF, _abi.IGX_HIM_BODIES, _abi.IGX_HIM_OBS
    gen = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=gen, device=device)  # noqa: E731
    lo, hi = (torch.tensor(v, device=device) for v in him_limits())
    root = torch.empty(n, 13, device=device)
    root[:, 0:2] = r(n, 2) * 0.5
    root[:, 2] = 0.9 + r(n) * 0.2
    root[:, 3:7] = _quats(n, gen, device)
    root[:, 7:13] = r(n, 6)
    dof_state = torch.empty(n, D, 2, device=device)
    dof_state[..., 0] = lo + (hi - lo) * (torch.rand(n, D, generator=gen, device=device) * 1.04 - 0.02)
    dof_state[..., 1] = r(n, D)
    init_root = torch.zeros(n, 13, device=device)
    init_root[:, 0:2] = r(n, 2) * 0.05
    init_root[:, 2] = 0.9
    init_root[:, 6] = 1.0
    frame = torch.cat([(lo + (hi - lo) * torch.rand(D, generator=gen, device=device)).double(),
                       (r(NB * 13) * 0.5).double()])
    t = dict(root_states=root, dof_state=dof_state, rigid_body_states=r(n, NB, 13) * 0.5,
             actions=torch.rand(n, D, generator=gen, device=device) * 2 - 1, initial_root_states=init_root,
             initial_dof_pos=torch.zeros(n, D, device=device), motion_frame=frame.contiguous(),
             obs=torch.zeros(n, NO, device=device), potentials=torch.full((n,), -1000.0 / 0.0166, device=device),
             prev_potentials=torch.full((n,), -1000.0 / 0.0166, device=device), rew=torch.zeros(n, device=device),
             reset=(torch.rand(n, generator=gen, device=device) < 0.02).to(torch.int64),
             progress=torch.randint(0, 1000, (n,), generator=gen, device=device),
             randomize=torch.zeros(n, dtype=torch.int64, device=device))
    if with_rand:
        t["rand"] = torch.rand(n, 2 * D, generator=gen, device=device)
    return t
