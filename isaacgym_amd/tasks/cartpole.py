"""Cartpole task: mirror of the reference's tasks/cartpole.py (class Cartpole)
on the fused HIP step.  `post_physics_step` is one kernel (progress, flagged
resets, observations, reward, reset flags; cartpole.py:170-178, :148-162,
:135-146, :185-201); `reset_idx(env_ids)` is the same reset on an id list."""
import torch

from .. import _abi, ops
from .base.vec_task import VecTask


class Cartpole(VecTask):
    REWARD_VARIANT = 0

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0):
        self.cfg = cfg
        self.reset_dist = self.cfg["env"]["resetDist"]
        self.max_push_effort = self.cfg["env"]["maxEffort"]
        self.max_episode_length = 500
        self.cfg["env"]["numObservations"] = 4
        self.cfg["env"]["numActions"] = 1
        super().__init__(self.cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        self.num_dof = 2
        self.dof_state = self.sim.acquire("dof_state")  # [N, 2, 2] interleaved (pos, vel)
        view = self.dof_state.view(self.num_envs, self.num_dof, 2)
        self.dof_pos = view[..., 0]
        self.dof_vel = view[..., 1]
        self.seed = int(seed)
        self.reset_rand = None  # optional [N,4] pre-drawn uniforms replacing the in-kernel generator
        self._draws = 0

    _PARAM_KEY = ("reset_dist", "max_episode_length", "seed")

    def _build_params(self):
        return _abi.CartpoleParams(reset_dist=self.reset_dist, max_episode_length_m1=float(self.max_episode_length - 1),
                                   reward_variant=self.REWARD_VARIANT, seed=self.seed)

    def _buffers(self):
        return dict(dof_state=self.dof_state, rand=self.reset_rand, obs=self.obs_buf, rew=self.rew_buf,
                    reset=self.reset_buf, progress=self.progress_buf)

    FUSED_ACTION_STAGE = True

    def pre_physics_step(self, actions):  # cartpole.py:164-168
        if not hasattr(self, "_forces"):
            self._forces = torch.zeros(self.num_envs * self.num_dof, device=self.device, dtype=torch.float)
        self._scale_actions(actions, self._forces, "first_dof", power_scale=self.max_push_effort, num_dof=self.num_dof)
        self.sim.apply_dof_forces(self._forces)

    def post_physics_step(self):
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail():  # the clamped copy and the time-out flags of VecTask.step in the same launch
            self._fuse_tail(p, b)
        ops.cartpole_step(p, b, self.num_envs, self._call_cache)

    def reset_idx(self, env_ids):
        ops.cartpole_reset(self._params(), self._buffers(), env_ids.to(self.device, torch.int64).contiguous())
        self.sim.notify_dof_state(env_ids)
