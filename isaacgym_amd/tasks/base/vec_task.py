"""VecTask: the vectorised-environment base class the tasks derive from.

Mirror of the reference's tasks/base/vec_task.py (classes Env + VecTask) for
everything the step path touches: the buffer contract of `allocate_buffers`
(:310-333), `step` (:370-418), `reset` (:436-449), `reset_done` (:451-466),
`get_state` (:353-355), `zero_actions` (:420-428) and the read-only properties.
Attribute names and dtypes are unchanged, so code written against the
reference's task objects keeps working.  What differs:

* physics is a `sim` object handed to the constructor (isaacgym_amd.sim), not
  an isaacgym handle; viewer, cameras and domain randomisation of simulation
  parameters have no counterpart here (`apply_randomizations` covers the
  observation / action noise, :630-729);
* the action noise and clamp, the time-out flags, the observation noise and the
  clamped observation copy of `step` are HIP kernels (igx_action_stage_step /
  igx_clamp, igx_vec_task_tail_step) writing into persistent tensors: the
  returned observation tensor is overwritten by the next `step` / `reset` call,
  where the reference returns a fresh tensor (cfg env.freshStepOutputs: true
  makes step() return copies instead).  The per-call noise draw is made
  inside the kernels (Philox), not by torch.randn_like.
"""
import numpy as np
import ctypes as C

import torch

from ... import ops, randomization


class VecTask:
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 24}
    # True for tasks whose pre_physics_step goes through _scale_actions: step() then leaves the
    # action noise and clamp to that same launch
    FUSED_ACTION_STAGE = False

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
        self._tdev = torch.device(self.device)
        # rl_device == sim device (the default): what step() returns needs no .to(); checked once, not per tensor per step
        self._rl_same = torch.device(rl_device) == self._tdev
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
        # freshStepOutputs: step() hands back copies (as the reference's fresh tensors) instead of the persistent
        # buffers that the next step overwrites - for callers that keep observations across steps
        self.fresh_step_outputs = bool(config["env"].get("freshStepOutputs", False))
        self.total_train_env_frames = 0
        self.control_steps = 0
        self.dt = config["sim"]["dt"]
        self.dr_randomizations = {}
        self.first_randomization = True
        self.last_step = -1
        self.last_rand_step = -1
        self.noise_seed = int(config.get("seed", 0) or 0)
        # recorded draws to use instead of the generators, for reproducing a torch run:
        # {"actions": [fresh per step, ...], "actions_corr": [corr per re-randomisation, ...], "observations": ...}
        self.replay_noise = None
        self._stage = None
        self._tail_done = False   # set by a post_physics_step whose kernel already wrote obs_out / time_outs

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

    def get_number_of_agents(self):  # what the learner-side wrapper asks its environment (utils/rlgames_utils.py:257-258)
        return self.num_agents

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
    # ---- parameter structs ----------------------------------------------------------
    _PARAM_KEY = ()  # names of the attributes _build_params() reads

    def _build_params(self):
        raise NotImplementedError

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        cls._PARAM_KEY_SET = frozenset(cls._PARAM_KEY)

    _PARAM_KEY_SET = frozenset()

    def __setattr__(self, name, value):
        # assigning one of the attributes the parameter struct is built from drops the kept struct
        if name in self._PARAM_KEY_SET:
            object.__setattr__(self, "_ptmpl", None)
        object.__setattr__(self, name, value)

    def _params(self):
        """This step's parameter struct.  Filling a ctypes struct with array members costs tens of
        microseconds, so the filled struct is kept and copied until one of the attributes it was
        built from is assigned again (the reference reads them every step: changing one still takes effect)."""
        self._draws += 1
        if self.__dict__.get("_ptmpl") is None:
            object.__setattr__(self, "_ptmpl", self._build_params())
        p = type(self._ptmpl)()
        C.memmove(C.byref(p), C.byref(self._ptmpl), C.sizeof(p))
        p.rng_offset = self._draws
        p.rng_offset_dev = self._rng_dev()
        if self.env_offset and hasattr(p, "env_offset"):
            p.env_offset = self.env_offset
        return p

    # index of this task's first environment in the whole (multi-GPU) job: with it, and one seed for all
    # shards, the in-kernel draws of a shard are those of the unsharded run (distributed.env_shard gives it)
    env_offset = 0

    @property
    def _call_cache(self):
        """Validated argument structs of this task's step calls, kept between steps (isaacgym_amd._lib.call_step)."""
        c = self.__dict__.get("_call_cache_dict")
        if c is None:
            c = self.__dict__["_call_cache_dict"] = {}
        return c

    def _rng_dev(self):
        """Address of the device-resident draw counter, or None.  isaacgym_amd.graph.GraphedStep installs the
        counter: a replayed graph repeats the by-value rng_offset, the counter keeps the draws fresh."""
        counter = getattr(self, "_rng_counter", None)
        return None if counter is None else counter.data_ptr()

    def pre_physics_step(self, actions):
        raise NotImplementedError

    def post_physics_step(self):
        raise NotImplementedError

    def reset_idx(self, env_ids):
        pass

    # ---- observation / action noise (vec_task.py:630-729) ------------------------
    def apply_randomizations(self, dr_params):
        """The non-physical part of the reference's method: when to re-randomise, the
        randomize_buf bookkeeping, and the parameters of the two noise lambdas."""
        rand_freq = dr_params.get("frequency", 1)
        self.last_step = self.sim.num_simulate_calls
        if self.first_randomization:
            do_nonenv_randomize = True
        else:
            do_nonenv_randomize = (self.last_step - self.last_rand_step) >= rand_freq
            rand_envs = torch.logical_and(self.randomize_buf >= rand_freq, self.reset_buf)
            self.randomize_buf[rand_envs] = 0
        if do_nonenv_randomize:
            self.last_rand_step = self.last_step
        made = []
        for name in randomization.NONPHYSICAL:
            if name in dr_params and do_nonenv_randomize:
                entry = randomization.noise_params(dr_params[name], self.last_step)
                # a fresh dict: the correlated draw of the previous one is dropped, as at :703 / :729
                entry["noise_lambda"] = self._make_noise_lambda(name)
                self.dr_randomizations[name] = entry
                made.append(entry)
        randomization.bind_operations(made)
        self.first_randomization = False

    def _noise_spec(self, name, like):
        """The NoiseSpec of dr_randomizations[name]; its correlated draw is made at first use (:695-698)."""
        entry = self.dr_randomizations[name]
        spec = entry.get("spec")
        if spec is None or spec.corr.numel() != like.numel():
            cs, csh, sc, sh = randomization.noise_coefficients(entry)
            corr = self._replayed(name + "_corr")
            if corr is None:
                corr = torch.randn_like(like, dtype=torch.float32)
            spec = ops.NoiseSpec(corr, cs, csh, sc, sh, entry["operation"],
                                 entry["distribution"], seed=(self.noise_seed << 1) | (name == "actions"))
            spec.offset = self.control_steps << 1
            spec.offset_dev = self._rng_dev()
            entry["spec"] = spec
            entry["corr"] = spec.corr
        return spec

    def _replayed(self, key):
        queue = self.replay_noise.get(key) if self.replay_noise else None
        return queue.pop(0).to(self.device, torch.float32).contiguous() if queue else None

    def _make_noise_lambda(self, name):
        def noise_lambda(tensor):
            x = tensor.to(self.device, torch.float32).contiguous()
            return ops.noise_apply(self._noise_spec(name, x), x, fresh=self._replayed(name))
        return noise_lambda

    def _noise_stage(self, name, like):
        """noise= / fresh= keyword arguments for the kernels, empty without that randomisation."""
        if not self.dr_randomizations.get(name, None):
            return {}
        return dict(noise=self._noise_spec(name, like), fresh=self._replayed(name))

    # ---- stepping (vec_task.py:370-418) ------------------------------------------
    def _stage_actions(self, actions):
        """Noise (:381-382) and clamp (:384) of the actions into self._action_tensor."""
        stage = self._noise_stage("actions", actions)
        if stage:
            actions = ops.noise_apply(stage["noise"], actions, fresh=stage["fresh"])
        ops.clamp(actions, -self.clip_actions, self.clip_actions, self._action_tensor)
        return self._action_tensor

    def _scale_actions(self, actions, targets, mode, gain=None, offset=None, power_scale=1.0, num_dof=0, keep=None):
        """pre_physics_step of the tasks that only scale their actions: copy into `keep`
        (self.actions) and scaling into `targets` in one launch, which also carries the action
        noise and clamp when the call comes from step()."""
        stage, self._stage = (self._stage or {}), None
        if actions.device != self._tdev or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(self.device, torch.float32).contiguous()
        ops.action_stage(actions, targets, mode, gain=gain, offset=offset, power_scale=power_scale, num_dof=num_dof,
                         actions_out=keep, cache=self._call_cache, **stage)

    def step(self, actions):
        if actions.device != self._tdev or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(self.device, torch.float32).contiguous()
        if actions.numel() != self._action_tensor.numel():
            raise ValueError("expected %d x %d actions, got %s" % (self.num_envs, self.num_actions, tuple(actions.shape)))
        if self.FUSED_ACTION_STAGE:
            self._stage = dict(clip_actions=self.clip_actions, **self._noise_stage("actions", actions))
            self.pre_physics_step(actions)
        else:
            self.pre_physics_step(self._stage_actions(actions))

        for _ in range(self.control_freq_inv):
            self.sim.simulate()

        self._tail_done = False
        self.post_physics_step()
        self.control_steps += 1

        if not self._tail_done:
            stage = self._noise_stage("observations", self.obs_buf)
            ops.vec_task_tail(self.obs_buf, self._obs_out, self.clip_obs, self.progress_buf, self.reset_buf,
                              self._time_outs, self.max_episode_length,
                              self.states_buf if self.num_states > 0 else None,
                              self._states_out if self.num_states > 0 else None,
                              obs_noised=self.obs_buf if stage else None, cache=self._call_cache, **stage)
        self.timeout_buf = self._time_outs
        if self.fresh_step_outputs:
            dev = self.rl_device
            fresh = (lambda t: t.clone()) if self._rl_same else (lambda t: t.to(dev))
            self.extras["time_outs"] = fresh(self.timeout_buf)
            obs = {"obs": fresh(self._obs_out)}
            if self.num_states > 0:
                obs["states"] = fresh(self._states_out)
            return obs, fresh(self.rew_buf), fresh(self.reset_buf), self.extras
        if self._rl_same:
            self.extras["time_outs"] = self.timeout_buf
            self.obs_dict["obs"] = self._obs_out
            if self.num_states > 0:
                self.obs_dict["states"] = self._states_out
            return self.obs_dict, self.rew_buf, self.reset_buf, self.extras
        self.extras["time_outs"] = self.timeout_buf.to(self.rl_device)
        self.obs_dict["obs"] = self._obs_out.to(self.rl_device)
        if self.num_states > 0:
            self.obs_dict["states"] = self._states_out.to(self.rl_device)
        return self.obs_dict, self.rew_buf.to(self.rl_device), self.reset_buf.to(self.rl_device), self.extras

    def zero_actions(self):
        return torch.zeros([self.num_envs, self.num_actions], dtype=torch.float32, device=self.rl_device)

    def _can_fuse_tail(self, states_too=False):
        """The step kernel may write the clamped observation copy and the time-out flags itself when
        nothing sits between it and the tail: no observation noise, and no privileged states unless
        the kernel clamps those as well."""
        return (states_too or self.num_states == 0) and not self.dr_randomizations.get("observations", None)

    def _fuse_tail(self, params, buffers):
        """Hand the persistent tail outputs to a kernel that fills them (obs_clamped, states_clamped, time_outs)."""
        params.clip_obs = float(self.clip_obs)
        buffers["obs_clamped"], buffers["time_outs"] = self._obs_out, self._time_outs
        if self.num_states > 0:
            buffers["states_clamped"] = self._states_out
        self._tail_done = True

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
