"""HumanoidImitate: the fork's 63-dof pose-tracking humanoid (reference tasks/humanoid_imitate.py).
One fused kernel per step: flagged resets, the 205-wide observation and the tracking reward
against frame 41 of the motion file."""
import numpy as np
import torch

from .. import _abi, ops
from .base.vec_task import VecTask


class HumanoidImitate(VecTask):
    POSE_FRAME = 41  # humanoid_imitate.py:698, :708

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0, motion=None):
        self.cfg = cfg
        env = cfg["env"]
        self.randomize = cfg["task"]["randomize"]
        self.dof_vel_scale = env["dofVelocityScale"]
        self.power_scale = env["powerScale"]
        self.up_weight = env["upWeight"]
        self.actions_cost_scale = env["actionsCost"]
        self.death_cost = env["deathCost"]
        self.termination_height = env["terminationHeight"]
        self.termination_radius = env["terminationRadius"]
        self.max_episode_length = env["episodeLength"]
        env["numObservations"] = _abi.IGX_HIM_OBS
        env["numActions"] = _abi.IGX_HIM_DOF
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        dev, n = self.device, self.num_envs
        self.num_dof, self.num_bodies = _abi.IGX_HIM_DOF, _abi.IGX_HIM_BODIES
        if motion is None:
            motion = np.loadtxt(cfg["motionToImitate"])  # humanoid_imitate.py:103-106
        motion = np.ascontiguousarray(motion, dtype=np.float64)
        if motion.ndim != 2 or motion.shape[1] != self.num_dof + self.num_bodies * 13 or motion.shape[0] <= self.POSE_FRAME:
            raise ValueError("the motion to imitate must be [> %d frames, %d]" % (self.POSE_FRAME, self.num_dof + self.num_bodies * 13))
        self.motion_to_imitate = torch.from_numpy(motion).to(dev)
        self._frame = self.motion_to_imitate[self.POSE_FRAME].contiguous()
        asset = self.sim.asset
        self.dof_limits_lower = torch.tensor(asset["dof_limits_lower"], device=dev, dtype=torch.float)
        self.dof_limits_upper = torch.tensor(asset["dof_limits_upper"], device=dev, dtype=torch.float)
        self.motor_efforts = torch.tensor(asset.get("motor_efforts", [100.0] * self.num_dof), device=dev, dtype=torch.float)
        self.root_states = self.sim.acquire("root_states")
        self.initial_root_states = self.root_states.clone()
        self.initial_root_states[:, 7:13] = 0
        self.dof_state = self.sim.acquire("dof_state")
        view = self.dof_state.view(n, self.num_dof, 2)
        self.dof_pos, self.dof_vel = view[..., 0], view[..., 1]
        self.rigid_body_states = self.sim.acquire("rigid_body_states").view(n, -1)
        zero = torch.zeros((), device=dev)
        self.initial_dof_pos = torch.where(self.dof_limits_lower > zero, self.dof_limits_lower,
                                           torch.where(self.dof_limits_upper < zero, self.dof_limits_upper,
                                                       torch.zeros(n, self.num_dof, device=dev))).contiguous()
        self.targets = torch.tensor([1000.0, 0.0, 0.0], device=dev).repeat((n, 1))
        self.potentials = torch.tensor([-1000.0 / self.dt], device=dev).repeat(n)
        self.prev_potentials = self.potentials.clone()
        self.actions = torch.zeros(n, self.num_dof, device=dev)
        self._forces = torch.zeros(n, self.num_dof, device=dev)
        self.seed = int(seed)
        self.reset_rand = None
        self._draws = 0
        self._lo, self._hi = self.dof_limits_lower.tolist(), self.dof_limits_upper.tolist()

    _PARAM_KEY = ("dt", "dof_vel_scale", "up_weight", "actions_cost_scale", "termination_height", "termination_radius",
                  "max_episode_length", "seed")

    def _build_params(self):
        return _abi.HumanoidImitateParams(
            dt=self.dt, dof_vel_scale=self.dof_vel_scale, up_weight=self.up_weight,
            actions_cost_scale=self.actions_cost_scale, termination_height=self.termination_height,
            termination_radius_sq=float(np.float32(self.termination_radius ** 2)),
            max_episode_length_m1=float(self.max_episode_length - 1), targets=[1000.0, 0.0, 0.0],
            dof_limits_lower=self._lo, dof_limits_upper=self._hi, seed=self.seed)

    def _buffers(self):
        return dict(root_states=self.root_states, dof_state=self.dof_state,
                    rigid_body_states=self.rigid_body_states, actions=self.actions,
                    initial_root_states=self.initial_root_states, initial_dof_pos=self.initial_dof_pos,
                    rand=self.reset_rand, motion_frame=self._frame, obs=self.obs_buf, potentials=self.potentials,
                    prev_potentials=self.prev_potentials, rew=self.rew_buf, reset=self.reset_buf,
                    progress=self.progress_buf, randomize=self.randomize_buf)

    FUSED_ACTION_STAGE = True

    def pre_physics_step(self, actions):  # humanoid_imitate.py:627-632
        self._scale_actions(actions, self._forces, "force", gain=self.motor_efforts, power_scale=self.power_scale,
                            keep=self.actions)
        self.sim.apply_dof_forces(self._forces)

    def post_physics_step(self):
        # humanoid_imitate.py:581-582: reset_idx re-randomises, i.e. on steps where some env is reset
        if self.randomize and bool(self.reset_buf.any()):
            self.apply_randomizations(self.randomization_params)
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail():  # clamped copy and time-out flags of VecTask.step in the same launch
            self._fuse_tail(p, b)
        ops.humanoid_imitate_step(p, b, self.num_envs, self._call_cache)

    def reset_idx(self, env_ids):
        """reset_idx on an id list = the flagged reset of the step kernel: raise the flags; they are
        applied at the start of the next post_physics_step, where the reference applies them."""
        self.reset_buf[env_ids] = 1
