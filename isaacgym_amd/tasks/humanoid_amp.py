"""HumanoidAMP task: mirror of the reference's tasks/humanoid_amp.py (class
HumanoidAMP) over tasks/amp/humanoid_amp_base.py (class HumanoidAMPBase).

`post_physics_step()` is one HIP kernel (progress, 105-wide observation,
reward, compute_humanoid_reset, AMP history shift and current AMP observation);
`reset_idx(env_ids)` is one scatter / recompute kernel plus, for reference-state
initialisation, one fused motion-sampling kernel that writes the AMP history;
`fetch_amp_obs_demo(num_samples)` is one fused kernel on top of the host numpy
sampling (same generator, same draws as the reference).
"""
import os
from enum import Enum

import ctypes as C

import numpy as np
import torch

from .. import _abi, _lib, ops
from .amp.utils_amp.motion_lib import DOF_OFFSETS, MotionLib
from .base.vec_task import VecTask

NUM_OBS = 13 + 52 + 28 + 12  # [root_h, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos]
NUM_ACTIONS = 28
NUM_AMP_OBS_PER_STEP = NUM_OBS
KEY_BODY_NAMES = ["right_hand", "left_hand", "right_foot", "left_foot"]
BODY_NAMES = ["pelvis", "torso", "head", "right_upper_arm", "right_lower_arm", "right_hand", "left_upper_arm",
              "left_lower_arm", "left_hand", "right_thigh", "right_shin", "right_foot", "left_thigh", "left_shin",
              "left_foot"]


class HumanoidAMPBase(VecTask):
    def __init__(self, config, rl_device, sim_device, graphics_device_id=0, headless=True,
                 virtual_screen_capture=False, force_render=False, sim=None):
        self.cfg = config
        env = config["env"]
        self._pd_control = env["pdControl"]
        self.power_scale = env["powerScale"]
        self.randomize = config["task"]["randomize"]
        self.debug_viz = env["enableDebugVis"]
        self.camera_follow = env.get("cameraFollow", False)
        self.max_episode_length = env["episodeLength"]
        self._local_root_obs = env["localRootObs"]
        self._contact_bodies = env["contactBodies"]
        self._termination_height = env["terminationHeight"]
        self._enable_early_termination = env["enableEarlyTermination"]
        env["numObservations"] = self.get_obs_size()
        env["numActions"] = self.get_action_size()
        super().__init__(config, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        self.dt = self.control_freq_inv * config["sim"]["dt"]  # humanoid_amp_base.py:94-95
        dev, n = self.device, self.num_envs
        self.num_dof, self.num_bodies = _abi.IGX_AMP_NUM_DOF, _abi.IGX_AMP_NUM_BODIES
        self._root_states = self.sim.acquire("root_states")
        self._initial_root_states = self._root_states.clone()
        self._initial_root_states[:, 7:13] = 0
        self._dof_state = self.sim.acquire("dof_state")  # [N,28,2]
        view = self._dof_state.view(n, self.num_dof, 2)
        self._dof_pos, self._dof_vel = view[..., 0], view[..., 1]
        self._initial_dof_pos = torch.zeros(n, self.num_dof, device=dev)
        # right_shoulder_x / left_shoulder_x: first dof of joints 2 and 4 (:125-128)
        self._initial_dof_pos[:, DOF_OFFSETS[2]] = 0.5 * np.pi
        self._initial_dof_pos[:, DOF_OFFSETS[4]] = -0.5 * np.pi
        self._initial_dof_vel = torch.zeros(n, self.num_dof, device=dev)
        self._rigid_body_state = self.sim.acquire("rigid_body_state")
        rb = self._rigid_body_state.view(n, self.num_bodies, 13)
        self._rigid_body_pos, self._rigid_body_rot = rb[..., 0:3], rb[..., 3:7]
        self._rigid_body_vel, self._rigid_body_ang_vel = rb[..., 7:10], rb[..., 10:13]
        self._contact_forces = self.sim.acquire("contact_forces").view(n, self.num_bodies, 3)
        self._terminate_buf = torch.ones(n, device=dev, dtype=torch.long)
        self._key_body_ids = torch.tensor([BODY_NAMES.index(b) for b in KEY_BODY_NAMES], device=dev, dtype=torch.long)
        self._contact_body_ids = torch.tensor([BODY_NAMES.index(b) for b in self._contact_bodies], device=dev,
                                              dtype=torch.long)
        if len(self._contact_bodies) > 4:
            raise ValueError("at most 4 contact bodies")
        self.actions = torch.zeros(n, NUM_ACTIONS, device=dev)
        self._targets = torch.zeros(n, NUM_ACTIONS, device=dev)
        asset = self.sim.asset
        if "dof_limits_lower" in asset:
            self.dof_limits_lower = torch.tensor(asset["dof_limits_lower"], device=dev, dtype=torch.float)
            self.dof_limits_upper = torch.tensor(asset["dof_limits_upper"], device=dev, dtype=torch.float)
        else:
            self.dof_limits_lower = torch.full((self.num_dof,), -np.pi, device=dev)
            self.dof_limits_upper = torch.full((self.num_dof,), np.pi, device=dev)
        self.motor_efforts = torch.tensor(asset.get("motor_efforts", [100.0] * self.num_dof), device=dev)
        if self._pd_control:
            self._build_pd_action_offset_scale()
        self._cb = [BODY_NAMES.index(b) for b in self._contact_bodies]
        self._kb = [BODY_NAMES.index(b) for b in KEY_BODY_NAMES]
        self._num_amp_obs_steps = env.get("numAMPObsSteps", 2)
        self._amp_obs_buf = torch.zeros((n, self._num_amp_obs_steps, NUM_AMP_OBS_PER_STEP), device=dev,
                                        dtype=torch.float)

    def get_obs_size(self):
        return NUM_OBS

    def get_action_size(self):
        return NUM_ACTIONS

    def _build_pd_action_offset_scale(self):  # humanoid_amp_base.py:310-343
        lim_low = self.dof_limits_lower.cpu().numpy().copy()
        lim_high = self.dof_limits_upper.cpu().numpy().copy()
        for j in range(len(DOF_OFFSETS) - 1):
            off, size = DOF_OFFSETS[j], DOF_OFFSETS[j + 1] - DOF_OFFSETS[j]
            if size == 3:
                lim_low[off:off + size] = -np.pi
                lim_high[off:off + size] = np.pi
            elif size == 1:
                mid = 0.5 * (lim_high[off] + lim_low[off])
                scale = 0.7 * (lim_high[off] - lim_low[off])  # a bit beyond the joint limits
                lim_low[off], lim_high[off] = mid - scale, mid + scale
        self._pd_action_offset = torch.tensor(0.5 * (lim_high + lim_low), device=self.device, dtype=torch.float)
        self._pd_action_scale = torch.tensor(0.5 * (lim_high - lim_low), device=self.device, dtype=torch.float)

    def _action_to_pd_targets(self, action):  # :467-469
        return self._pd_action_offset + self._pd_action_scale * action

    FUSED_ACTION_STAGE = True

    def pre_physics_step(self, actions):  # :410-422
        if self._pd_control:
            self._scale_actions(actions, self._targets, "pd_target", gain=self._pd_action_scale,
                                offset=self._pd_action_offset, keep=self.actions)
            self.sim.apply_dof_position_targets(self._targets)
        else:
            self._scale_actions(actions, self._targets, "force", gain=self.motor_efforts, power_scale=self.power_scale,
                                keep=self.actions)
            self.sim.apply_dof_forces(self._targets)

    def _refresh_sim_tensors(self):
        self.sim.refresh()

    # ---- kernels ---------------------------------------------------------------------
    def _params(self):
        cb, kb = self._cb, self._kb
        return _abi.AmpParams(num_amp_obs_steps=self._num_amp_obs_steps, local_root_obs=int(bool(self._local_root_obs)),
                              enable_early_termination=int(bool(self._enable_early_termination)),
                              num_contact_bodies=len(cb), contact_body_ids=cb + [0] * (4 - len(cb)), key_body_ids=kb,
                              termination_height=self._termination_height,
                              max_episode_length_m1=float(self.max_episode_length - 1))

    def _buffers(self):
        return dict(root_states=self._root_states, dof_state=self._dof_state,
                    rigid_body_state=self._rigid_body_state, contact_forces=self._contact_forces,
                    obs=self.obs_buf, amp_obs=self._amp_obs_buf, rew=self.rew_buf, reset=self.reset_buf,
                    terminate=self._terminate_buf, progress=self.progress_buf)

    def post_physics_step(self):  # humanoid_amp_base.py:424-438 (+ the AMP history, humanoid_amp.py:312-330)
        self._refresh_sim_tensors()
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail():  # clamped copy and time-out flags of VecTask.step in the same launch
            self._fuse_tail(p, b)
        ops.humanoid_amp_step(p, b, self.num_envs, self._call_cache)
        self.extras["terminate"] = self._terminate_buf

    def reset_idx(self, env_ids):  # humanoid_amp_base.py:165-169, :392-408
        env_ids = env_ids.to(self.device, torch.int64).contiguous()
        state = dict(dof_pos=self._initial_dof_pos[env_ids].contiguous(),
                     dof_vel=self._initial_dof_vel[env_ids].contiguous())
        ops.humanoid_amp_reset(self._params(), self._buffers(), env_ids, state)
        self.sim.notify_root_state(self._initial_root_states, env_ids)
        self.sim.notify_dof_state(env_ids)


class HumanoidAMP(HumanoidAMPBase):
    class StateInit(Enum):
        Default = 0
        Start = 1
        Random = 2
        Hybrid = 3

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, motion_lib=None, allow_pickle=False):
        self.cfg = cfg
        self._state_init = HumanoidAMP.StateInit[cfg["env"]["stateInit"]]
        self._hybrid_init_prob = cfg["env"]["hybridInitProb"]
        self._num_amp_obs_steps = cfg["env"]["numAMPObsSteps"]
        assert self._num_amp_obs_steps >= 2
        self._reset_default_env_ids = []
        self._reset_ref_env_ids = []
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim)
        if motion_lib is not None:
            self._motion_lib = motion_lib
        else:
            motion_file = cfg["env"].get("motion_file", "amp_humanoid_backflip.npy")
            if not os.path.isabs(motion_file):
                motion_file = os.path.join(cfg["env"].get("motion_dir", "assets/amp/motions"), motion_file)
            self._load_motion(motion_file, allow_pickle)
        self.num_amp_obs = self._num_amp_obs_steps * NUM_AMP_OBS_PER_STEP
        self._curr_amp_obs_buf = self._amp_obs_buf[:, 0]
        self._hist_amp_obs_buf = self._amp_obs_buf[:, 1:]
        self._amp_obs_demo_buf = None

    def _load_motion(self, motion_file, allow_pickle=False):
        self._motion_lib = MotionLib(motion_file=motion_file, num_dofs=self.num_dof,
                                     key_body_ids=self._key_body_ids.cpu().numpy(), device=self.device,
                                     allow_pickle=allow_pickle)

    def get_num_amp_obs(self):
        return self.num_amp_obs

    @property
    def amp_observation_space(self):
        return dict(low=-np.inf, high=np.inf, shape=(self.num_amp_obs,))

    def post_physics_step(self):  # humanoid_amp.py:125-134
        super().post_physics_step()
        self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.get_num_amp_obs())

    def fetch_amp_obs_demo(self, num_samples):  # humanoid_amp.py:146-171
        dt = self.dt
        motion_ids = self._motion_lib.sample_motions(num_samples)
        if self._amp_obs_demo_buf is None:
            self._build_amp_obs_demo_buf(num_samples)
        else:
            assert self._amp_obs_demo_buf.shape[0] == num_samples
        motion_times0 = self._motion_lib.sample_time(motion_ids)
        # :156-159 tile the ids over the history steps and step the times back by dt; the kernel does that itself
        # (same fp64 operations), so the host sends num_samples ids / times instead of num_samples * steps
        self._motion_lib.get_amp_obs_steps(motion_ids, motion_times0, self._num_amp_obs_steps, dt, self._local_root_obs,
                                           self._amp_obs_demo_buf)
        return self._amp_obs_demo_buf.view(-1, self.get_num_amp_obs())

    def fetch_amp_obs_demo_device(self, num_samples, seed=0):
        """fetch_amp_obs_demo with its two draws made on the device, inside the launch that builds the batch
        (igx_motion_fetch_amp_obs; the draws are also left in self._demo_draws): nothing crosses the host link and
        nothing synchronises, a fetch is one launch.  The distributions are the reference's
        (motion by weight, uniform phase); the random stream is the library's Philox, keyed by (seed, fetch count,
        sample index), not numpy's - use fetch_amp_obs_demo to reproduce a numpy-seeded sequence."""
        if self._amp_obs_demo_buf is None:
            self._build_amp_obs_demo_buf(num_samples)
        else:
            assert self._amp_obs_demo_buf.shape[0] == num_samples
        w = self.__dict__.get("_demo_draws")
        if w is None or w[0].shape[0] != num_samples:
            w = self._demo_draws = (torch.empty(num_samples, dtype=torch.int64, device=self.device),
                                    torch.empty(num_samples, dtype=torch.float64, device=self.device))
        self._demo_fetches = self.__dict__.get("_demo_fetches", 0) + 1
        ops.motion_fetch_amp_obs(self._motion_lib._tables, self._motion_lib.sampling_cdf(), num_samples,
                                 self._num_amp_obs_steps, self.dt, self._local_root_obs, self._amp_obs_demo_buf,
                                 motion_ids=w[0], motion_times=w[1], seed=seed, rng_offset=self._demo_fetches)
        return self._amp_obs_demo_buf.view(-1, self.get_num_amp_obs())

    def stream_amp_obs_demo(self, recorder, num_samples, device_sampling=True, seed=0):
        """One fetch, handed to a whole-row TrajectoryRecorder (recorder.TrajectoryRecorder(num_samples,
        get_num_amp_obs(), None, max_fetches, device)): the batch goes to pinned host memory on the recorder's side
        stream while the next fetch is computed.  Returns the device batch (valid until the next fetch)."""
        batch = self.fetch_amp_obs_demo_device(num_samples, seed) if device_sampling else self.fetch_amp_obs_demo(num_samples)
        recorder.capture(batch)
        return batch

    def _build_amp_obs_demo_buf(self, num_samples):
        self._amp_obs_demo_buf = torch.zeros((num_samples, self._num_amp_obs_steps, NUM_AMP_OBS_PER_STEP),
                                             device=self.device, dtype=torch.float)

    # ---- resets (humanoid_amp.py:184-310) ------------------------------------------------
    def reset_idx(self, env_ids):
        env_ids = env_ids.to(self.device, torch.int64).contiguous()
        self._reset_actors(env_ids)
        self._init_amp_obs(env_ids)

    def _reset_actors(self, env_ids):
        if self._state_init == HumanoidAMP.StateInit.Default:
            self._reset_default(env_ids)
        elif self._state_init in (HumanoidAMP.StateInit.Start, HumanoidAMP.StateInit.Random):
            self._reset_ref_state_init(env_ids)
        elif self._state_init == HumanoidAMP.StateInit.Hybrid:
            self._reset_hybrid_state_init(env_ids)
        else:
            assert False, "Unsupported state initialization strategy: {:s}".format(str(self._state_init))

    def _reset_default(self, env_ids):
        # dof state <- initial pose; the root row is handed to the engine only (the
        # shared tensor keeps the old root until the next physics refresh, as in the reference)
        HumanoidAMPBase.reset_idx(self, env_ids)
        self._reset_default_env_ids = env_ids

    def _reset_ref_state_init(self, env_ids):
        num_envs = env_ids.shape[0]
        motion_ids = self._motion_lib.sample_motions(num_envs)
        if self._state_init in (HumanoidAMP.StateInit.Random, HumanoidAMP.StateInit.Hybrid):
            motion_times = self._motion_lib.sample_time(motion_ids)
        elif self._state_init == HumanoidAMP.StateInit.Start:
            motion_times = np.zeros(num_envs)
        else:
            assert False, "Unsupported state initialization strategy: {:s}".format(str(self._state_init))
        root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos = \
            self._motion_lib.get_motion_state(motion_ids, motion_times)
        self._set_env_state(env_ids=env_ids, root_pos=root_pos, root_rot=root_rot, dof_pos=dof_pos,
                            root_vel=root_vel, root_ang_vel=root_ang_vel, dof_vel=dof_vel)
        self._reset_ref_env_ids = env_ids
        self._reset_ref_motion_ids = motion_ids
        self._reset_ref_motion_times = motion_times

    def _reset_hybrid_state_init(self, env_ids):
        num_envs = env_ids.shape[0]
        ref_probs = torch.tensor(np.array([self._hybrid_init_prob] * num_envs), device=self.device, dtype=torch.float)
        ref_init_mask = torch.bernoulli(ref_probs) == 1.0
        ref_reset_ids = env_ids[ref_init_mask]
        if len(ref_reset_ids) > 0:
            self._reset_ref_state_init(ref_reset_ids.contiguous())
        default_reset_ids = env_ids[torch.logical_not(ref_init_mask)]
        if len(default_reset_ids) > 0:
            self._reset_default(default_reset_ids.contiguous())

    def reset_flagged(self, seed=0):
        """reset_done() without the host: the flagged environments are found, the choice between reference
        state and default pose made (stateInit Hybrid), reference motions and times drawn (weights and uniform
        phase as MotionLib.sample_motions / sample_time; the library's counter-based generator stands where
        numpy's and torch.bernoulli stand), the motion states evaluated and written, and the AMP history
        filled, all by kernels that run over every environment and skip the ones without the flag.
        No nonzero(), no device-to-host copy, no numpy."""
        SI = HumanoidAMP.StateInit
        ref_prob = {SI.Default: 0.0, SI.Start: 1.0, SI.Random: 1.0, SI.Hybrid: float(self._hybrid_init_prob)}[self._state_init]
        n, K, dev, lib = self.num_envs, self._num_amp_obs_steps, self.device, self._motion_lib
        w = getattr(self, "_flagged_ws", None)
        if w is None:
            f32 = dict(dtype=torch.float32, device=dev)
            w = self._flagged_ws = dict(
                env_ids=torch.empty(n, dtype=torch.int64, device=dev), ids=torch.empty(n, dtype=torch.int64, device=dev),
                default_ids=torch.empty(n, dtype=torch.int64, device=dev),
                times=torch.empty(n, dtype=torch.float64, device=dev),
                state=dict(root_pos=torch.empty(n, 3, **f32), root_rot=torch.empty(n, 4, **f32),
                           dof_pos=torch.empty(n, 28, **f32), root_vel=torch.empty(n, 3, **f32),
                           root_ang_vel=torch.empty(n, 3, **f32), dof_vel=torch.empty(n, 28, **f32)),
                draws=0)
        w["draws"] += 1
        if self._run_flagged_plan(w, seed, ref_prob):
            return self._publish_obs()
        ops.motion_sample_flagged(lib._tables, lib.sampling_cdf(), self.reset_buf, w["env_ids"], w["ids"], w["times"],
                                  random_time=self._state_init in (SI.Random, SI.Hybrid), seed=seed,
                                  rng_offset=w["draws"], rng_offset_dev=self._rng_dev(), ref_init_prob=ref_prob, env_offset=self.env_offset,
                                  default_env_ids=w["default_ids"])
        if ref_prob < 1.0:  # default pose: joints from the initial state, the root row stays (humanoid_amp_base.py:392-408)
            ops.humanoid_amp_reset(self._params(), self._buffers(), w["default_ids"],
                                   dict(dof_pos=self._initial_dof_pos, dof_vel=self._initial_dof_vel))
        if ref_prob <= 0.0:
            self.sim.notify_dof_state(None)
            self._build_flagged_plan(w, seed, ref_prob)
            return self._publish_obs()
        ops.motion_state(lib._tables, w["ids"], w["times"], w["state"])
        ops.humanoid_amp_reset(self._params(), self._buffers(), w["env_ids"], w["state"])
        # the history slots: the same motions at times - dt, - 2 dt, ... (humanoid_amp.py:279-294), tiled in the kernel
        ops.motion_amp_obs_history(lib._tables, w["ids"], w["times"], None, K, self.dt, self._local_root_obs, self._amp_obs_buf)
        self.sim.notify_root_state(self._root_states, None)
        self.sim.notify_dof_state(None)
        self._build_flagged_plan(w, seed, ref_prob)
        return self._publish_obs()

    def _build_flagged_plan(self, w, seed, ref_prob):
        """After a first reset_flagged() went through the checked calls: the argument structs of the same launches,
        kept for the calls that follow (the buffers are this task's own, and the plan is dropped as soon as one of
        them points elsewhere).  A later call then costs the launches and one field store."""
        SI = HumanoidAMP.StateInit
        lib, ml, n, K = _lib.lib(), self._motion_lib, self.num_envs, self._num_amp_obs_steps
        sampler = _abi.MotionSampler(cdf=_lib.tensor_ptr(ml.sampling_cdf(), torch.float64),
                                     random_time=int(self._state_init in (SI.Random, SI.Hybrid)), seed=int(seed),
                                     rng_offset=0, rng_offset_dev=self._rng_dev(), ref_init_prob=float(ref_prob),
                                     env_offset=int(self.env_offset))
        default_state = dict(dof_pos=self._initial_dof_pos, dof_vel=self._initial_dof_vel)
        bufs = self._buffers()
        amp_b, dev = _lib.make_buffers(_abi.AmpBuffers, bufs)
        st_ref, _ = _lib.make_buffers(_abi.MotionStateOut, w["state"])
        st_def, _ = _lib.make_buffers(_abi.MotionStateOut, default_state)
        held = _lib._held([ml.sampling_cdf(), self.reset_buf, w["env_ids"], w["ids"], w["times"], w["default_ids"]] +
                          list(bufs.values()) + list(w["state"].values()) + list(default_state.values()))
        ptr = lambda t, dt: _lib.tensor_ptr(t, dt)  # noqa: E731
        i64, f64, f32, cn = torch.int64, torch.float64, torch.float32, C.c_int64(n)
        params = self._params()
        calls = [(lib.igx_motion_sample_flagged, (C.byref(ml._tables), C.byref(sampler), ptr(self.reset_buf, i64), cn,
                                                  ptr(w["env_ids"], i64), ptr(w["ids"], i64), ptr(w["times"], f64),
                                                  ptr(w["default_ids"], i64)))]
        if ref_prob < 1.0:
            calls.append((lib.igx_humanoid_amp_reset, (C.byref(params), C.byref(amp_b), ptr(w["default_ids"], i64), cn,
                                                       C.byref(st_def))))
        if ref_prob > 0.0:
            calls.append((lib.igx_motion_state, (C.byref(ml._tables), ptr(w["ids"], i64), ptr(w["times"], f64), cn,
                                                 C.byref(st_ref))))
            calls.append((lib.igx_humanoid_amp_reset, (C.byref(params), C.byref(amp_b), ptr(w["env_ids"], i64), cn,
                                                       C.byref(st_ref))))
            calls.append((lib.igx_motion_amp_obs_history, (C.byref(ml._tables), ptr(w["ids"], i64), ptr(w["times"], f64), None,
                                                           cn, C.c_int32(K), C.c_int32(1), C.c_double(float(self.dt)),
                                                           C.c_int32(int(bool(self._local_root_obs))),
                                                           ptr(self._amp_obs_buf, f32))))
        w["plan"] = dict(key=self._flagged_plan_key(seed, ref_prob), calls=calls, sampler=sampler, dev=dev, held=held,
                         ptrs=_lib._pointers(held), storages=_lib._storages(held), keep=(params, amp_b, st_ref, st_def))

    def _flagged_plan_key(self, seed, ref_prob):
        """Everything the plan's structs were filled from: the tensor objects behind the buffer names (a name bound
        to another tensor is another buffer; where each object points is compared separately) and the parameters."""
        return (tuple(map(id, self._buffers().values())), id(self._initial_dof_pos), id(self._initial_dof_vel),
                id(self._motion_lib), id(self._motion_lib._tables), id(self._motion_lib.sampling_cdf()),
                int(seed), float(ref_prob), self._state_init, self._rng_dev(), self.env_offset, self._num_amp_obs_steps,
                float(self.dt), bool(self._local_root_obs), tuple(self._kb), tuple(self._cb),
                bool(self._enable_early_termination), float(self._termination_height), int(self.max_episode_length))

    def _run_flagged_plan(self, w, seed, ref_prob):
        plan = w.get("plan")
        if plan is None:
            return False
        if plan["key"] != self._flagged_plan_key(seed, ref_prob) or _lib._pointers(plan["held"]) != plan["ptrs"]:
            w["plan"] = None  # something the structs were built from has changed: back through the checked calls
            return False
        plan["sampler"].rng_offset = w["draws"]
        dev = plan["dev"]
        for fn, args in plan["calls"]:
            _lib.check(_lib._launch(fn, dev, *args))
        if ref_prob > 0.0:
            self.sim.notify_root_state(self._root_states, None)
        self.sim.notify_dof_state(None)
        return True

    def _set_env_state(self, env_ids, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel):
        state = dict(root_pos=root_pos, root_rot=root_rot, dof_pos=dof_pos, root_vel=root_vel,
                     root_ang_vel=root_ang_vel, dof_vel=dof_vel)
        ops.humanoid_amp_reset(self._params(), self._buffers(), env_ids, state)
        self.sim.notify_root_state(self._root_states, env_ids)
        self.sim.notify_dof_state(env_ids)

    def _init_amp_obs(self, env_ids):
        # the reset kernel already wrote obs / the current AMP slot of env_ids and copied it into
        # their history; what is left are the two (never cleared, humanoid_amp.py:263-272) id lists
        if len(self._reset_default_env_ids) > 0:
            self._init_amp_obs_default(self._reset_default_env_ids)
        if len(self._reset_ref_env_ids) > 0:
            self._init_amp_obs_ref(self._reset_ref_env_ids, self._reset_ref_motion_ids, self._reset_ref_motion_times)

    def _init_amp_obs_default(self, env_ids):
        self._hist_amp_obs_buf[env_ids] = self._curr_amp_obs_buf[env_ids].unsqueeze(-2)

    def _init_amp_obs_ref(self, env_ids, motion_ids, motion_times):  # humanoid_amp.py:279-294
        # the np.tile of the ids and `times - dt * (arange(K - 1) + 1)` of the reference happen in the kernel: the host
        # sends one id and one time per reset environment
        ids, times = self._motion_lib._to_device(motion_ids, motion_times)
        ops.motion_amp_obs_history(self._motion_lib._tables, ids, times, env_ids, self._num_amp_obs_steps, self.dt,
                                   self._local_root_obs, self._amp_obs_buf)
