"""Shared host side of the Ant and Humanoid tasks (the reference keeps two
near-identical classes, tasks/ant.py and tasks/humanoid.py)."""
import torch

from .. import _abi, ops
from .base.vec_task import VecTask


class LocomotionTask(VecTask):
    NUM_OBS = None
    NUM_ACTIONS = None
    NUM_SENSOR_FLOATS = None
    HUMANOID = False

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0):
        self.cfg = cfg
        env = cfg["env"]
        self.max_episode_length = env["episodeLength"]
        self.randomization_params = cfg["task"].get("randomization_params", {})
        self.randomize = cfg["task"]["randomize"]
        self.dof_vel_scale = env["dofVelocityScale"]
        self.angular_velocity_scale = env.get("angularVelocityScale", 0.1)
        self.contact_force_scale = env["contactForceScale"]
        self.power_scale = env["powerScale"]
        self.heading_weight = env["headingWeight"]
        self.up_weight = env["upWeight"]
        self.actions_cost_scale = env["actionsCost"]
        self.energy_cost_scale = env["energyCost"]
        self.joints_at_limit_cost_scale = env["jointsAtLimitCost"]
        self.death_cost = env["deathCost"]
        self.termination_height = env["terminationHeight"]
        self.debug_viz = env["enableDebugVis"]
        env["numObservations"] = self.NUM_OBS
        env["numActions"] = self.NUM_ACTIONS
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        if self.randomize:  # create_sim: ant.py:143-144, humanoid.py:154-155
            self.apply_randomizations(self.randomization_params)
        dev, n = self.device, self.num_envs
        self.num_dof = self.NUM_ACTIONS
        asset = self.sim.asset
        self.dof_limits_lower = torch.tensor(asset["dof_limits_lower"], device=dev, dtype=torch.float)
        self.dof_limits_upper = torch.tensor(asset["dof_limits_upper"], device=dev, dtype=torch.float)
        self.motor_efforts = torch.tensor(asset.get("motor_efforts", [1.0] * self.num_dof), device=dev, dtype=torch.float)
        self.joint_gears = self.motor_efforts
        self.max_motor_effort = float(self.motor_efforts.max()) if self.HUMANOID else 0.0

        self.root_states = self.sim.acquire("root_states")                  # [N,13]
        self.initial_root_states = self.root_states.clone()
        self.initial_root_states[:, 7:13] = 0
        self.dof_state = self.sim.acquire("dof_state")                      # [N,D,2]
        view = self.dof_state.view(n, self.num_dof, 2)
        self.dof_pos, self.dof_vel = view[..., 0], view[..., 1]
        self.vec_sensor_tensor = self.sim.acquire("sensors").view(n, self.NUM_SENSOR_FLOATS)
        if self.HUMANOID:
            self.dof_force_tensor = self.sim.acquire("dof_force").view(n, self.num_dof)
        zero = torch.zeros((), device=dev)
        self.initial_dof_pos = torch.zeros(n, self.num_dof, device=dev)
        self.initial_dof_pos = torch.where(self.dof_limits_lower > zero, self.dof_limits_lower,
                                           torch.where(self.dof_limits_upper < zero, self.dof_limits_upper,
                                                       self.initial_dof_pos)).contiguous()
        self.initial_dof_vel = torch.zeros(n, self.num_dof, device=dev)

        self.up_axis_idx = 2
        self.up_vec = torch.tensor([0.0, 0.0, 1.0], device=dev).repeat((n, 1))
        self.heading_vec = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat((n, 1))
        self.start_rotation = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)
        self.inv_start_rot = torch.tensor([-0.0, -0.0, -0.0, 1.0], device=dev).repeat((n, 1))
        self.basis_vec0 = self.heading_vec.clone()
        self.basis_vec1 = self.up_vec.clone()
        self.targets = torch.tensor([1000.0, 0.0, 0.0], device=dev).repeat((n, 1))
        self.target_dirs = torch.tensor([1.0, 0.0, 0.0], device=dev).repeat((n, 1))
        self.potentials = torch.tensor([-1000.0 / self.dt], device=dev).repeat(n)
        self.prev_potentials = self.potentials.clone()
        self.actions = torch.zeros(n, self.num_dof, device=dev)
        self._forces = torch.zeros(n, self.num_dof, device=dev)
        self.seed = int(seed)
        self.reset_rand = None  # optional [N, 2*num_dof] pre-drawn uniforms
        self._draws = 0
        self._const = None
        self._true_objective = torch.zeros(n, device=dev)

    # the per-env constant tensors of the reference (targets, inv_start_rot, basis
    # vectors) are folded into the parameter block: one value for all envs
    _PARAM_KEY = ("dt", "dof_vel_scale", "contact_force_scale", "angular_velocity_scale", "up_weight", "heading_weight",
                  "actions_cost_scale", "energy_cost_scale", "joints_at_limit_cost_scale", "termination_height",
                  "death_cost", "max_episode_length", "seed")

    def _build_params(self):
        if self._const is None:  # one device-to-host read, at the first call
            self._const = dict(targets=self.targets[0].tolist(), inv_start_rot=self.inv_start_rot[0].tolist(),
                               basis_vec0=self.basis_vec0[0].tolist(), basis_vec1=self.basis_vec1[0].tolist(),
                               dof_limits_lower=self.dof_limits_lower.tolist(),
                               dof_limits_upper=self.dof_limits_upper.tolist())
            if self.HUMANOID:
                self._const.update(motor_efforts=self.motor_efforts.tolist(), max_motor_effort=self.max_motor_effort)
        return _abi.LocoParams(
            num_dof=self.num_dof, num_sensor_floats=self.NUM_SENSOR_FLOATS, dt=self.dt,
            dof_vel_scale=self.dof_vel_scale, contact_force_scale=self.contact_force_scale,
            angular_velocity_scale=self.angular_velocity_scale, up_weight=self.up_weight,
            heading_weight=self.heading_weight, actions_cost_scale=self.actions_cost_scale,
            energy_cost_scale=self.energy_cost_scale, joints_at_limit_cost_scale=self.joints_at_limit_cost_scale,
            termination_height=self.termination_height, death_cost=self.death_cost,
            max_episode_length_m1=float(self.max_episode_length - 1), seed=self.seed,
            **self._const)

    def _buffers(self):
        b = dict(root_states=self.root_states, dof_state=self.dof_state, sensors=self.vec_sensor_tensor,
                 actions=self.actions, initial_root_states=self.initial_root_states,
                 initial_dof_pos=self.initial_dof_pos, rand=self.reset_rand, obs=self.obs_buf,
                 potentials=self.potentials, prev_potentials=self.prev_potentials, up_vec=self.up_vec,
                 heading_vec=self.heading_vec, rew=self.rew_buf, reset=self.reset_buf, progress=self.progress_buf,
                 randomize=self.randomize_buf)
        if self.HUMANOID:
            b["dof_force"] = self.dof_force_tensor
        else:
            b["true_objective"] = self._true_objective
        return b

    FUSED_ACTION_STAGE = True

    def pre_physics_step(self, actions):  # ant.py:368-374, humanoid.py:392-396
        self._scale_actions(actions, self._forces, "force", gain=self.joint_gears, power_scale=self.power_scale,
                            keep=self.actions)
        self.sim.apply_dof_forces(self._forces)

    def post_physics_step(self):
        # ant.py:341-342, humanoid.py:362-363: reset_idx re-randomises, i.e. on steps where some env is reset
        if self.randomize and bool(self.reset_buf.any()):
            self.apply_randomizations(self.randomization_params)
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail():  # the clamped copy and the time-out flags of VecTask.step in the same launch
            self._fuse_tail(p, b)
        self._step_kernel(p, b, self.num_envs, self._call_cache)
        if not self.HUMANOID:
            self.extras["true_objective"] = self._true_objective  # ant.py:332-337

    def reset_idx(self, env_ids):
        if self.randomize:  # ant.py:341-342, humanoid.py:362-363: every reset_idx re-randomises first
            self.apply_randomizations(self.randomization_params)
        env_ids = env_ids.to(self.device, torch.int64).contiguous()
        self._reset_kernel(self._params(), self._buffers(), env_ids)
        self.sim.notify_root_state(self.initial_root_states, env_ids)
        self.sim.notify_dof_state(env_ids)


class Ant(LocomotionTask):
    """tasks/ant.py: 60 observations, 8 actions, 4 force sensors x 6 floats."""
    NUM_OBS, NUM_ACTIONS, NUM_SENSOR_FLOATS, HUMANOID = 60, 8, 24, False
    _step_kernel = staticmethod(ops.ant_step)
    _reset_kernel = staticmethod(ops.ant_reset)


class Humanoid(LocomotionTask):
    """tasks/humanoid.py: 108 observations, 21 actions, 2 force sensors x 6 floats."""
    NUM_OBS, NUM_ACTIONS, NUM_SENSOR_FLOATS, HUMANOID = 108, 21, 12, True
    _step_kernel = staticmethod(ops.humanoid_step)
    _reset_kernel = staticmethod(ops.humanoid_reset)
