"""VecTask: the vectorised-environment base class the tasks derive from.

Mirror of the reference's tasks/base/vec_task.py (classes Env + VecTask) for
everything the step path touches: the buffer contract of `allocate_buffers`
(:310-333), `step` (:370-418), `reset` (:436-449), `reset_done` (:451-466),
`get_state` (:353-355), `zero_actions` (:420-428) and the read-only properties.
Attribute names and dtypes are unchanged, so code written against the
reference's task objects keeps working.  What differs:

* physics is a `sim` object handed to the constructor (isaacgym_amd.sim), not
  an isaacgym handle; viewer, cameras and domain randomisation of simulation
  parameters have no counterpart here;
* the action clamp, the time-out flags and the clamped observation copy of
  `step` are HIP kernels (igx_clamp, igx_vec_task_tail_step) writing into
  persistent tensors: the returned observation tensor is overwritten by the
  next `step` / `reset` call, where the reference returns a fresh tensor.
"""
import numpy as np
import torch

from ... import ops


class VecTask:
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 24}

    def __init__(self, config, rl_device, sim_device, graphics_device_id=0, headless=True,
                 virtual_screen_capture=False, force_render=False, sim=None):
        if sim is None:
            raise ValueError("a sim object is required (see isaacgym_amd.sim.SyntheticSim / ReplaySim)")
        self.cfg = config
        split_device = str(sim_device).split(":")
        self.device_type = split_device[0]
        self.device_id = int(split_device[1]) if len(split_device) > 1 else 0
        if self.device_type.lower() not in ("cuda", "gpu"):
            raise ValueError("isaacgym_amd runs on a GPU only (sim_device=%r); there is no CPU pipeline" % sim_device)
        self.device = "cuda:%d" % self.device_id
        self.rl_device = rl_device
        self.headless = headless
        self.graphics_device_id = graphics_device_id
        self.virtual_screen_capture = virtual_screen_capture
        self.force_render = force_render
        self.viewer = None

        env = config["env"]
        self.num_environments = env["numEnvs"]
        self.num_agents = env.get("numAgents", 1)
        self.num_observations = env.get("numObservations", 0)
        self.num_states = env.get("numStates", 0)
        self.num_actions = env["numActions"]
        self.control_freq_inv = env.get("controlFrequencyInv", 1)
        self.clip_obs = env.get("clipObservations", np.inf)
        self.clip_actions = env.get("clipActions", np.inf)
        self.total_train_env_frames = 0
        self.control_steps = 0
        self.dt = config["sim"]["dt"]
        self.dr_randomizations = {}
        self.first_randomization = True

        self.sim = sim
        if sim.num_envs != self.num_environments:
            raise ValueError("sim has %d envs, config %d" % (sim.num_envs, self.num_environments))
        if torch.device(self.device) != sim.device:
            raise ValueError("sim lives on %s, task on %s" % (sim.device, self.device))
        self.sim_initialized = True
        self.allocate_buffers()
        self.obs_dict = {}

    # ---- reference properties ------------------------------------------------
    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_acts(self):
        return self.num_actions

    @property
    def num_obs(self):
        return self.num_observations

    @property
    def observation_space(self):
        return dict(low=-np.inf, high=np.inf, shape=(self.num_obs,))

    @property
    def state_space(self):
        return dict(low=-np.inf, high=np.inf, shape=(self.num_states,))

    @property
    def action_space(self):
        return dict(low=-1.0, high=1.0, shape=(self.num_actions,))

    def set_train_info(self, env_frames, *args, **kwargs):
        self.total_train_env_frames = env_frames

    def get_env_state(self):
        return None

    def set_env_state(self, env_state):
        pass

    # ---- buffers (vec_task.py:310-333) ----------------------------------------
    def allocate_buffers(self):
        n, dev = self.num_envs, self.device
        self.obs_buf = torch.zeros((n, self.num_obs), device=dev, dtype=torch.float)
        self.states_buf = torch.zeros((n, self.num_states), device=dev, dtype=torch.float)
        self.rew_buf = torch.zeros(n, device=dev, dtype=torch.float)
        self.reset_buf = torch.ones(n, device=dev, dtype=torch.long)
        self.timeout_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.progress_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.randomize_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.extras = {}
        # persistent outputs of the step tail
        self._obs_out = torch.zeros_like(self.obs_buf)
        self._states_out = torch.zeros_like(self.states_buf)
        self._time_outs = torch.zeros(n, device=dev, dtype=torch.bool)
        self._action_tensor = torch.zeros((n, self.num_actions), device=dev, dtype=torch.float)

    def get_state(self):
        """clamp(states_buf, +-clip_obs) (vec_task.py:353-355)."""
        ops.clamp(self.states_buf, -self.clip_obs, self.clip_obs, self._states_out)
        return self._states_out.to(self.rl_device)

    # ---- to be provided by the task -------------------------------------------
    def pre_physics_step(self, actions):
        raise NotImplementedError

    def post_physics_step(self):
        raise NotImplementedError

    def reset_idx(self, env_ids):
        pass

    # ---- stepping (vec_task.py:370-418) ------------------------------------------
    def step(self, actions):
        if self.dr_randomizations.get("actions", None):
            actions = self.dr_randomizations["actions"]["noise_lambda"](actions)
        actions = actions.to(self.device, torch.float32).contiguous()
        if actions.numel() != self._action_tensor.numel():
            raise ValueError("expected %d x %d actions, got %s" % (self.num_envs, self.num_actions, tuple(actions.shape)))
        ops.clamp(actions, -self.clip_actions, self.clip_actions, self._action_tensor)
        self.pre_physics_step(self._action_tensor)

        for _ in range(self.control_freq_inv):
            self.sim.simulate()

        self.post_physics_step()
        self.control_steps += 1

        if self.dr_randomizations.get("observations", None):
            self.obs_buf = self.dr_randomizations["observations"]["noise_lambda"](self.obs_buf).contiguous()
        ops.vec_task_tail(self.obs_buf, self._obs_out, self.clip_obs, self.progress_buf, self.reset_buf,
                          self._time_outs, self.max_episode_length,
                          self.states_buf if self.num_states > 0 else None,
                          self._states_out if self.num_states > 0 else None)
        self.timeout_buf = self._time_outs
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = self._obs_out.to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self._states_out.to(self.rl_device)
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def zero_actions(self):
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def _publish_obs(self):
        ops.clamp(self.obs_buf, -self.clip_obs, self.clip_obs, self._obs_out)
        self.obs_dict["obs"] = self._obs_out.to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self.get_state()
        return self.obs_dict

    def reset(self):
        """Called once at start; does not compute observations (vec_task.py:436-449)."""
        return self._publish_obs()

    def reset_done(self):
        """reset_idx on every flagged environment (vec_task.py:451-466)."""
        done_env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(done_env_ids) > 0:
            self.reset_idx(done_env_ids)
        return self._publish_obs(), done_env_ids

    def render(self, mode="rgb_array"):
        return None
