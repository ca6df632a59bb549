"""Low-level Python entry points: one function per C-ABI call of libigx.so.

`params` is the matching isaacgym_amd._abi.*Params struct, `tensors` maps the
field names of the *Buffers struct to contiguous device tensors.
"""
import ctypes as C

import torch

from . import _abi, _lib


def cartpole_step(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_cartpole_step", params, _abi.CartpoleBuffers, tensors, n_envs, cache)


def ant_step(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_ant_step", params, _abi.LocoBuffers, tensors, n_envs, cache)


def humanoid_step(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_humanoid_step", params, _abi.LocoBuffers, tensors, n_envs, cache)


def cartpole_reset(params, tensors, env_ids):
    _lib.call_reset("igx_cartpole_reset", params, _abi.CartpoleBuffers, tensors, env_ids)


def ant_reset(params, tensors, env_ids):
    _lib.call_reset("igx_ant_reset", params, _abi.LocoBuffers, tensors, env_ids)


def humanoid_reset(params, tensors, env_ids):
    _lib.call_reset("igx_humanoid_reset", params, _abi.LocoBuffers, tensors, env_ids)


def humanoid_imitate_step(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_humanoid_imitate_step", params, _abi.HumanoidImitateBuffers, tensors, n_envs, cache)


def allegro_pre_physics(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_allegro_pre_physics", params, _abi.AllegroBuffers, tensors, n_envs, cache)


def allegro_post_physics(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_allegro_post_physics", params, _abi.AllegroBuffers, tensors, n_envs, cache)


def allegro_static_prepare(device):
    """Resolve the library-owned scratch of allegro_post_physics_static on `device` (once per device, outside
    any graph capture); the step call can then be captured."""
    with torch.cuda.device(device):
        _lib.check(_lib.lib().igx_allegro_static_prepare())


def allegro_post_physics_static(params, tensors, n_envs, cache=None):
    """igx_allegro_post_physics with the reduction records in the library's own file-scope device
    array: allocates nothing and ignores tensors['workspace'] (n_envs <= IGX_ALLEGRO_STATIC_MAX_ENVS)."""
    _lib.call_step("igx_allegro_post_physics_static", params, _abi.AllegroBuffers, tensors, n_envs, cache)


# ---- AMP motion library / HumanoidAMP ---------------------------------------
_TABLE_DTYPES = dict(motion_length=torch.float64, motion_dt=torch.float64, motion_num_frames=torch.int64,
                     frame_offset=torch.int64)


def allegro_post_physics_shard(params, tensors, n_envs, cache=None):
    """One environment shard: leaves the 16-byte {num_resets, sum} record in tensors['workspace']."""
    _lib.call_step("igx_allegro_post_physics_shard", params, _abi.AllegroBuffers, tensors, n_envs, cache)


def allegro_finalize_shards(params, tensors, records, num_steps=1):
    """records: contiguous device tensor of num_shards x num_steps 16-byte records, gathered in rank order
    (shard-major); num_steps > 1 advances consecutive_successes over a queue of steps, in step order."""
    _lib.check_sizes(_abi.AllegroBuffers, params, {"consecutive_successes": tensors.get("consecutive_successes")}, 0,
                     "igx_allegro_finalize_shards")
    b, dev = _lib.make_buffers(_abi.AllegroBuffers, tensors)
    nbytes = records.numel() * records.element_size()
    num_steps = int(num_steps)
    if num_steps < 1 or nbytes == 0 or nbytes % (16 * num_steps):
        raise ValueError("records must hold a whole number of shards x %d 16-byte records" % num_steps)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_allegro_finalize_shards_queued(C.byref(params), C.byref(b), _lib.tensor_ptr(records),
                                                       C.c_int32(nbytes // (16 * num_steps)), C.c_int32(num_steps),
                                                       _lib.current_stream(dev))
    _lib.check(rc)


def hand_force_constants(force_decay, dt, force_decay_interval, force_prob_range):
    """(decay, log_prob_lo, log_prob_hi) as the reference's float32 torch expressions give them
    (tasks/allegro_hand.py:191-194, :558, :618): evaluated once on the host with the same torch CPU ops."""
    decay = float(torch.pow(torch.tensor(float(force_decay), dtype=torch.float32), dt / force_decay_interval))
    rng = torch.tensor([float(force_prob_range[0]), float(force_prob_range[1])], dtype=torch.float32)
    logs = torch.log(rng)
    return decay, float(logs[0]), float(logs[1])


def hand_random_forces(rb_forces, random_force_prob, reset, object_rb_handle, object_rb_mass, force_scale, decay,
                       log_prob_lo, log_prob_hi, rand=None, seed=0, rng_offset=0, rng_offset_dev=None, env_offset=0):
    """igx_hand_random_forces on rb_forces [N, B, 3]: zero + re-draw for flagged environments, decay, new object
    forces where the Bernoulli draw says so.  Call before the task's pre-physics kernel of the same step."""
    if rb_forces.dim() != 3 or rb_forces.shape[2] != 3:
        raise ValueError("rb_forces must be [num_envs, num_bodies, 3]")
    n, nb = rb_forces.shape[0], rb_forces.shape[1]
    if random_force_prob.numel() < n or reset.numel() < n:
        raise ValueError("random_force_prob and reset must have one entry per environment")
    if rand is not None and rand.numel() < n * _abi.IGX_HAND_FORCE_RAND_PER_ENV:
        raise ValueError("rand must be [num_envs, %d]" % _abi.IGX_HAND_FORCE_RAND_PER_ENV)
    dev = rb_forces.device
    f = torch.float32
    s = _abi.HandForces(rb_forces=_lib.tensor_ptr(rb_forces, f), random_force_prob=_lib.tensor_ptr(random_force_prob, f),
                        reset=_lib.tensor_ptr(reset, torch.int64), rand=_lib.tensor_ptr(rand, f) if rand is not None else None,
                        num_bodies=int(nb), object_rb_handle=int(object_rb_handle), object_rb_mass=float(object_rb_mass),
                        force_scale=float(force_scale), decay=float(decay), log_prob_lo=float(log_prob_lo),
                        log_prob_hi=float(log_prob_hi), seed=int(seed), rng_offset=int(rng_offset),
                        rng_offset_dev=rng_offset_dev, env_offset=int(env_offset))
    for t in (random_force_prob, reset) + ((rand,) if rand is not None else ()):
        if t.device != dev:
            raise ValueError("all tensors must live on %s" % dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_hand_random_forces(C.byref(s), C.c_int64(n), _lib.current_stream(dev))
    _lib.check(rc)


# ---- ShadowHand ------------------------------------------------------------------------------
def shadow_pre_physics(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_shadow_pre_physics", params, _abi.ShadowBuffers, tensors, n_envs, cache)


def shadow_post_physics(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_shadow_post_physics", params, _abi.ShadowBuffers, tensors, n_envs, cache)


def shadow_post_physics_shard(params, tensors, n_envs, cache=None):
    """One environment shard: leaves the 16-byte {num_resets, sum} record in tensors['workspace']."""
    _lib.call_step("igx_shadow_post_physics_shard", params, _abi.ShadowBuffers, tensors, n_envs, cache)


def shadow_finalize_shards(params, tensors, records, num_steps=1):
    """records: contiguous device tensor of num_shards x num_steps 16-byte records, gathered in rank order
    (shard-major); num_steps > 1 advances consecutive_successes over a queue of steps, in step order."""
    _lib.check_sizes(_abi.ShadowBuffers, params, {"consecutive_successes": tensors.get("consecutive_successes")}, 0,
                     "igx_shadow_finalize_shards")
    b, dev = _lib.make_buffers(_abi.ShadowBuffers, tensors)
    nbytes = records.numel() * records.element_size()
    num_steps = int(num_steps)
    if num_steps < 1 or nbytes == 0 or nbytes % (16 * num_steps):
        raise ValueError("records must hold a whole number of shards x %d 16-byte records" % num_steps)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_shadow_finalize_shards_queued(C.byref(params), C.byref(b), _lib.tensor_ptr(records),
                                                       C.c_int32(nbytes // (16 * num_steps)), C.c_int32(num_steps),
                                                       _lib.current_stream(dev))
    _lib.check(rc)


def motion_tables(tensors):
    """igx_motion_tables from name -> device tensor.  The struct holds raw device addresses; the tensors they came
    from ride along on the returned object (`_tensors`), so they live as long as it does."""
    t = _abi.MotionTables()
    t._tensors = dict(tensors)
    t.num_motions = int(tensors["motion_length"].shape[0])
    for name, ctype in _abi.MotionTables._fields_:
        if ctype is _abi.ptr:
            setattr(t, name, _lib.tensor_ptr(tensors[name], _TABLE_DTYPES.get(name, torch.float32)))
    return t


def motion_state(tables, motion_ids, motion_times, out):
    """out: name -> preallocated device tensor for any subset of igx_motion_state_out."""
    n = motion_ids.shape[0]
    if motion_times.shape[0] != n:
        raise ValueError("igx_motion_state: motion_ids and motion_times differ in length")
    widths = dict(root_pos=3, root_rot=4, dof_pos=28, root_vel=3, root_ang_vel=3, dof_vel=28, key_pos=12,
                  frame_idx0=1, frame_idx1=1, blend=1)
    for k, t in out.items():
        if t is not None and t.numel() < n * widths.get(k, 0):
            raise ValueError("igx_motion_state: out[%r] holds %d elements, %d samples need %d" %
                             (k, t.numel(), n, n * widths[k]))
    o, _ = _lib.make_buffers(_abi.MotionStateOut, out)
    dev = motion_ids.device
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_state(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                         _lib.tensor_ptr(motion_times, torch.float64),
                                         C.c_int64(motion_ids.shape[0]), C.byref(o), _lib.current_stream(dev))
    _lib.check(rc)


def motion_sample_flagged(tables, cdf, reset, env_ids, motion_ids, motion_times, random_time=True, seed=0, rng_offset=0,
                          rng_offset_dev=None, ref_init_prob=1.0, default_env_ids=None, env_offset=0):
    """Reference-state sampling for the environments whose reset flag is set, on the device (no nonzero(),
    no numpy): fills env_ids / motion_ids / motion_times [n]; environments without the flag get -1 / -1 / 0,
    which motion_state, motion_amp_obs and humanoid_amp_reset skip.  With ref_init_prob < 1 a flagged
    environment takes the default pose with probability 1 - ref_init_prob: it is listed in default_env_ids
    instead (stateInit Hybrid; 0 = stateInit Default)."""
    dev = reset.device
    s = _abi.MotionSampler(cdf=_lib.tensor_ptr(cdf, torch.float64), random_time=int(bool(random_time)), seed=int(seed),
                           rng_offset=int(rng_offset), rng_offset_dev=rng_offset_dev, ref_init_prob=float(ref_init_prob),
                           env_offset=int(env_offset))
    n = reset.shape[0]
    for t in (env_ids, motion_ids, motion_times) + ((default_env_ids,) if default_env_ids is not None else ()):
        if t.shape[0] != n:
            raise ValueError("outputs must have one entry per environment")
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_sample_flagged(C.byref(tables), C.byref(s), _lib.tensor_ptr(reset, torch.int64),
                                                  C.c_int64(n), _lib.tensor_ptr(env_ids, torch.int64),
                                                  _lib.tensor_ptr(motion_ids, torch.int64),
                                                  _lib.tensor_ptr(motion_times, torch.float64),
                                                  _lib.tensor_ptr(default_env_ids, torch.int64), _lib.current_stream(dev))
    _lib.check(rc)


def motion_amp_obs(tables, motion_ids, motion_times, local_root_obs, out, out_rows=None, out_row_stride=None,
                   out_col_offset=0):
    dev = motion_ids.device
    stride = _abi.IGX_AMP_OBS if out_row_stride is None else out_row_stride
    n = motion_ids.shape[0]
    if motion_times.shape[0] != n or (out_rows is not None and out_rows.shape[0] != n):
        raise ValueError("igx_motion_amp_obs: motion_ids, motion_times and out_rows differ in length")
    if out_col_offset < 0 or out_col_offset + _abi.IGX_AMP_OBS > stride:
        raise ValueError("igx_motion_amp_obs: a %d-float observation at column %d does not fit a row of %d" %
                         (_abi.IGX_AMP_OBS, out_col_offset, stride))
    if out_rows is None and out.numel() < n * stride:
        raise ValueError("igx_motion_amp_obs: out holds %d floats, %d samples need %d" % (out.numel(), n, n * stride))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_amp_obs(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                           _lib.tensor_ptr(motion_times, torch.float64),
                                           C.c_int64(motion_ids.shape[0]), C.c_int32(int(bool(local_root_obs))),
                                           _lib.tensor_ptr(out, torch.float32),
                                           _lib.tensor_ptr(out_rows, torch.int64), C.c_int64(stride),
                                           C.c_int64(out_col_offset), _lib.current_stream(dev))
    _lib.check(rc)


def motion_amp_obs_steps(tables, motion_ids, motion_times, num_steps, step_dt, local_root_obs, out):
    """fetch_amp_obs_demo after its draws, in one launch: out[i, k*105:(k+1)*105] = observation of motion
    motion_ids[i] at motion_times[i] + (-step_dt * k), k < num_steps (out: [n, num_steps * 105], contiguous)."""
    dev = motion_ids.device
    n = motion_ids.shape[0]
    if motion_times.shape[0] != n:
        raise ValueError("igx_motion_amp_obs_steps: motion_ids and motion_times differ in length")
    if not out.is_contiguous() or out.numel() < n * num_steps * _abi.IGX_AMP_OBS:
        raise ValueError("igx_motion_amp_obs_steps: out must be contiguous and hold %d x %d x %d floats" %
                         (n, num_steps, _abi.IGX_AMP_OBS))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_amp_obs_steps(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                                 _lib.tensor_ptr(motion_times, torch.float64), C.c_int64(n),
                                                 C.c_int32(int(num_steps)), C.c_double(float(step_dt)),
                                                 C.c_int32(int(bool(local_root_obs))), _lib.tensor_ptr(out, torch.float32),
                                                 _lib.current_stream(dev))
    _lib.check(rc)


def motion_sample(tables, cdf, motion_ids, motion_times, random_time=True, seed=0, rng_offset=0, rng_offset_dev=None,
                  env_offset=0):
    """MotionLib.sample_motions + sample_time for every entry of motion_ids / motion_times [n], on the device
    (the library's Philox stream; the distributions are the reference's)."""
    dev = motion_ids.device
    n = motion_ids.shape[0]
    if motion_times.shape[0] != n:
        raise ValueError("igx_motion_sample: motion_ids and motion_times differ in length")
    s = _abi.MotionSampler(cdf=_lib.tensor_ptr(cdf, torch.float64), random_time=int(bool(random_time)), seed=int(seed),
                           rng_offset=int(rng_offset), rng_offset_dev=rng_offset_dev, ref_init_prob=1.0,
                           env_offset=int(env_offset))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_sample(C.byref(tables), C.byref(s), C.c_int64(n), _lib.tensor_ptr(motion_ids, torch.int64),
                                          _lib.tensor_ptr(motion_times, torch.float64), _lib.current_stream(dev))
    _lib.check(rc)


def motion_amp_obs_history(tables, motion_ids, motion_times, env_ids, num_steps, step_dt, local_root_obs, amp_obs,
                           first_step=1):
    """History slots first_step ... num_steps - 1 of amp_obs [num_envs, num_steps, 105] for the listed environments
    (env_ids None: entry i is environment i) from their reference motions at motion_times - step_dt * k; entries
    whose motion id is negative are skipped."""
    dev = amp_obs.device
    n = motion_ids.shape[0]
    if motion_times.shape[0] != n or (env_ids is not None and env_ids.shape[0] != n):
        raise ValueError("igx_motion_amp_obs_history: the lists differ in length")
    if not amp_obs.is_contiguous() or amp_obs.numel() % (num_steps * _abi.IGX_AMP_OBS):
        raise ValueError("igx_motion_amp_obs_history: amp_obs must be a contiguous [num_envs, %d, %d] buffer" %
                         (num_steps, _abi.IGX_AMP_OBS))
    if env_ids is None and amp_obs.numel() < n * num_steps * _abi.IGX_AMP_OBS:
        raise ValueError("igx_motion_amp_obs_history: amp_obs holds fewer environments than the lists")
    if _lib.DEBUG_CHECKS and env_ids is not None and n:  # synchronises: the ids are device memory
        envs = amp_obs.numel() // (num_steps * _abi.IGX_AMP_OBS)
        if int(env_ids.max()) >= envs:
            raise ValueError("igx_motion_amp_obs_history: env_ids reach %d, amp_obs holds %d environments" %
                             (int(env_ids.max()), envs))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_amp_obs_history(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                                   _lib.tensor_ptr(motion_times, torch.float64),
                                                   _lib.tensor_ptr(env_ids, torch.int64), C.c_int64(n),
                                                   C.c_int32(int(num_steps)), C.c_int32(int(first_step)),
                                                   C.c_double(float(step_dt)), C.c_int32(int(bool(local_root_obs))),
                                                   _lib.tensor_ptr(amp_obs, torch.float32), _lib.current_stream(dev))
    _lib.check(rc)


def motion_fetch_amp_obs(tables, cdf, num_samples, num_steps, step_dt, local_root_obs, out, motion_ids=None,
                         motion_times=None, random_time=True, seed=0, rng_offset=0, rng_offset_dev=None, env_offset=0):
    """A whole fetch_amp_obs_demo in one launch: the draws of motion_sample and the batch of motion_amp_obs_steps on
    them (bit for bit what the two calls give).  out: [num_samples, num_steps * 105]; motion_ids / motion_times, when
    given, receive the draws."""
    dev = out.device
    n = int(num_samples)
    if not out.is_contiguous() or out.numel() < n * num_steps * _abi.IGX_AMP_OBS:
        raise ValueError("igx_motion_fetch_amp_obs: out must be contiguous and hold %d x %d x %d floats" %
                         (n, num_steps, _abi.IGX_AMP_OBS))
    for t in (motion_ids, motion_times):
        if t is not None and t.shape[0] < n:
            raise ValueError("igx_motion_fetch_amp_obs: motion_ids / motion_times shorter than num_samples")
    s = _abi.MotionSampler(cdf=_lib.tensor_ptr(cdf, torch.float64), random_time=int(bool(random_time)), seed=int(seed),
                           rng_offset=int(rng_offset), rng_offset_dev=rng_offset_dev, ref_init_prob=1.0,
                           env_offset=int(env_offset))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_fetch_amp_obs(C.byref(tables), C.byref(s), C.c_int64(n), C.c_int32(int(num_steps)),
                                                 C.c_double(float(step_dt)), C.c_int32(int(bool(local_root_obs))),
                                                 _lib.tensor_ptr(out, torch.float32), _lib.tensor_ptr(motion_ids, torch.int64),
                                                 _lib.tensor_ptr(motion_times, torch.float64), _lib.current_stream(dev))
    _lib.check(rc)


def amp_observations(root_states, dof_pos, dof_vel, key_body_pos, local_root_obs, obs):
    dev = obs.device
    f = torch.float32
    n = obs.shape[0]
    for name, t, w in (("root_states", root_states, 13), ("dof_pos", dof_pos, 28), ("dof_vel", dof_vel, 28),
                       ("key_body_pos", key_body_pos, 12), ("obs", obs, _abi.IGX_AMP_OBS)):
        if t.numel() < n * w:
            raise ValueError("igx_amp_observations: %s holds %d floats, %d rows need %d" % (name, t.numel(), n, n * w))
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_amp_observations(_lib.tensor_ptr(root_states, f), _lib.tensor_ptr(dof_pos, f),
                                             _lib.tensor_ptr(dof_vel, f), _lib.tensor_ptr(key_body_pos, f),
                                             C.c_int32(int(bool(local_root_obs))), _lib.tensor_ptr(obs, f),
                                             C.c_int64(obs.shape[0]), _lib.current_stream(dev))
    _lib.check(rc)


def humanoid_amp_step(params, tensors, n_envs, cache=None):
    _lib.call_step("igx_humanoid_amp_step", params, _abi.AmpBuffers, tensors, n_envs, cache)


def humanoid_amp_reset(params, tensors, env_ids, state):
    """state: root_pos / root_rot / dof_pos / root_vel / root_ang_vel / dof_vel rows for env_ids."""
    _lib.check_env_ids(_abi.AmpBuffers, params, tensors, env_ids, "igx_humanoid_amp_reset")
    m = env_ids.shape[0]
    for k, w in (("root_pos", 3), ("root_rot", 4), ("dof_pos", 28), ("root_vel", 3), ("root_ang_vel", 3), ("dof_vel", 28)):
        t = state.get(k)  # a missing entry leaves that part of the environment's state as it is
        if t is not None and t.numel() < m * w:
            raise ValueError("igx_humanoid_amp_reset: state[%r] must hold %d x %d floats" % (k, m, w))
    b, dev = _lib.make_buffers(_abi.AmpBuffers, tensors)
    st, _ = _lib.make_buffers(_abi.MotionStateOut, state)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_humanoid_amp_reset(C.byref(params), C.byref(b), _lib.tensor_ptr(env_ids, torch.int64),
                                               C.c_int64(env_ids.shape[0]), C.byref(st), _lib.current_stream(dev))
    _lib.check(rc)


# ---- VecTask.step tail --------------------------------------------------------
class NoiseSpec:
    """One entry of VecTask.dr_randomizations as the kernels take it (igx_noise):
    noise = ((corr * corr_scale + corr_shift) + fresh * scale) + shift, added to or
    multiplied with the tensor.  `corr` is the persistent standard-normal draw; the
    per-call draw is either handed in (`fresh=`) or made in the kernel from
    (seed, offset), offset advancing by one per use."""

    def __init__(self, corr, corr_scale, corr_shift, scale, shift, operation="additive", distribution="gaussian", seed=0):
        if operation not in ("additive", "scaling"):
            raise ValueError("operation must be 'additive' or 'scaling', got %r" % (operation,))
        if distribution not in ("gaussian", "uniform"):
            raise ValueError("distribution must be 'gaussian' or 'uniform', got %r" % (distribution,))
        self.corr = corr
        self.corr_scale, self.corr_shift, self.scale, self.shift = corr_scale, corr_shift, scale, shift
        self.operation, self.distribution = operation, distribution
        self.seed, self.offset = int(seed), 0
        self.offset_dev = None  # address of a device-resident counter added to offset (replayed graphs)

    def struct(self, count, fresh=None):
        if self.corr.numel() != count or (fresh is not None and fresh.numel() != count):
            raise ValueError("noise tensors must have %d elements" % count)
        nz = _abi.Noise(corr=_lib.tensor_ptr(self.corr, torch.float32),
                        fresh=_lib.tensor_ptr(fresh, torch.float32) if fresh is not None else None,
                        corr_scale=self.corr_scale, corr_shift=self.corr_shift, scale=self.scale, shift=self.shift,
                        operation=_abi.IGX_NOISE_SCALING if self.operation == "scaling" else _abi.IGX_NOISE_ADDITIVE,
                        distribution=_abi.IGX_NOISE_UNIFORM if self.distribution == "uniform" else _abi.IGX_NOISE_GAUSSIAN,
                        seed=self.seed, offset=self.offset, offset_dev=self.offset_dev)
        if fresh is None:
            self.offset += 1
        return nz


def noise_apply(spec, x, out=None, fresh=None):
    """out = noise_lambda(x) (tasks/base/vec_task.py:693-701, :720-727)."""
    dev = x.device
    out = torch.empty_like(x) if out is None else out
    nz = spec.struct(x.numel(), fresh)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_noise_apply(C.byref(nz), _lib.tensor_ptr(x, torch.float32), _lib.tensor_ptr(out, torch.float32),
                                        C.c_int64(x.numel()), _lib.current_stream(dev))
    _lib.check(rc)
    return out


def action_stage(actions, targets, mode, gain=None, offset=None, power_scale=1.0, clip_actions=float("inf"),
                 actions_out=None, noise=None, fresh=None, num_dof=0, cache=None):
    """Noise, clamp, copy and scaling of the actions in one launch (igx_action_stage_step).
    mode: "force" (a * gain * power_scale), "pd_target" (offset + gain * a) or
    "first_dof" (targets[i, 0] = a[i] * power_scale, targets [N, num_dof]).
    `cache`: a dict owned by the caller; without noise the checked argument struct is reused while the
    same tensors and scalars come back."""
    f = torch.float32
    dev = actions.device
    key = None
    if cache is not None and noise is None:
        key = (id(actions), id(targets), id(actions_out), id(gain), id(offset), mode, power_scale, clip_actions, num_dof)
        ent = cache.get("igx_action_stage_step")
        if ent is not None and ent[0] == key and _lib._pointers(ent[4]) == ent[5]:
            _lib.check(_lib._launch(_lib.lib().igx_action_stage_step, dev, ent[1], ent[2]))
            return
    modes = {"force": _abi.IGX_ACT_FORCE, "pd_target": _abi.IGX_ACT_PD_TARGET, "first_dof": _abi.IGX_ACT_FIRST_DOF}
    if mode not in modes:
        raise ValueError("mode must be one of %s" % sorted(modes))
    if mode == "first_dof":
        n, a = actions.numel(), 1
        if targets.numel() != n * num_dof:
            raise ValueError("targets must hold num_envs * num_dof floats")
    else:
        a = gain.numel()
        n = actions.numel() // a
        if n * a != actions.numel() or targets.numel() != actions.numel():
            raise ValueError("actions and targets must be [N, %d]" % a)
        if mode == "pd_target" and (offset is None or offset.numel() != a):
            raise ValueError("pd_target mode needs an offset per action")
    if actions_out is not None and actions_out.numel() != actions.numel():
        raise ValueError("actions_out must have as many elements as actions")
    s = _abi.ActionStage(actions=_lib.tensor_ptr(actions, f), targets=_lib.tensor_ptr(targets, f),
                         actions_out=_lib.tensor_ptr(actions_out, f) if actions_out is not None else None,
                         gain=_lib.tensor_ptr(gain, f) if gain is not None else None,
                         offset=_lib.tensor_ptr(offset, f) if offset is not None else None,
                         clip_actions=float(clip_actions), power_scale=float(power_scale), mode=modes[mode],
                         num_actions=a, num_dof=int(num_dof))
    if noise is not None:
        s.noise = noise.struct(actions.numel(), fresh)
    if key is not None:
        held = _lib._held((actions, targets, actions_out, gain, offset))
        cache["igx_action_stage_step"] = (key, C.byref(s), C.c_int64(n), s, held, _lib._pointers(held), _lib._storages(held))
    _lib.check(_lib._launch(_lib.lib().igx_action_stage_step, dev, C.byref(s), C.c_int64(n)))


def vec_task_tail(obs_buf, obs_out, clip_obs, progress, reset, time_outs, max_episode_length, states_buf=None,
                  states_out=None, noise=None, fresh=None, obs_noised=None, cache=None):
    """obs_out = clamp(obs_buf, +-clip_obs); states likewise; time_outs (torch.bool) =
    (progress >= max_episode_length - 1) & (reset != 0).  With `noise` (a NoiseSpec) the
    observation noise is applied before the clamp and the noised buffer kept in
    `obs_noised` (may be obs_buf itself)."""
    f = torch.float32
    dev = obs_buf.device
    key = None
    if cache is not None and noise is None:  # without noise nothing in the struct changes between steps
        key = (id(obs_buf), id(obs_out), id(progress), id(reset), id(time_outs), id(states_buf), id(states_out),
               clip_obs, max_episode_length)
        ent = cache.get("igx_vec_task_tail_step")
        if ent is not None and ent[0] == key and _lib._pointers(ent[4]) == ent[5]:
            _lib.check(_lib._launch(_lib.lib().igx_vec_task_tail_step, dev, ent[1], ent[2]))
            return
    t = _abi.VecTaskTail()
    t.obs_buf, t.obs_out, t.obs_count = _lib.tensor_ptr(obs_buf, f), _lib.tensor_ptr(obs_out, f), obs_buf.numel()
    if obs_out.numel() != obs_buf.numel():
        raise ValueError("obs_out must have as many elements as obs_buf")
    if noise is not None:
        t.obs_noise = noise.struct(obs_buf.numel(), fresh)
        if obs_noised is not None:
            if obs_noised.numel() != obs_buf.numel():
                raise ValueError("obs_noised must have as many elements as obs_buf")
            t.obs_noised = _lib.tensor_ptr(obs_noised, f)
    if states_buf is not None and states_buf.numel():
        if states_out is None or states_out.numel() != states_buf.numel():
            raise ValueError("states_out must have as many elements as states_buf")
        t.states_buf, t.states_out = _lib.tensor_ptr(states_buf, f), _lib.tensor_ptr(states_out, f)
        t.states_count = states_buf.numel()
    n = progress.shape[0]
    if reset.shape[0] != n or time_outs.shape[0] != n:
        raise ValueError("progress, reset and time_outs must have one entry per environment")
    t.progress, t.reset = _lib.tensor_ptr(progress, torch.int64), _lib.tensor_ptr(reset, torch.int64)
    t.time_outs = _lib.tensor_ptr(time_outs, torch.bool)
    t.max_episode_length_m1 = float(max_episode_length - 1)
    t.clip_obs = float(clip_obs)
    if key is not None:
        held = _lib._held((obs_buf, obs_out, progress, reset, time_outs, states_buf, states_out))
        cache["igx_vec_task_tail_step"] = (key, C.byref(t), C.c_int64(n), t, held, _lib._pointers(held), _lib._storages(held))
    _lib.check(_lib._launch(_lib.lib().igx_vec_task_tail_step, dev, C.byref(t), C.c_int64(n)))


def clamp(x, lo, hi, out):
    dev = x.device
    if out.numel() != x.numel():
        raise ValueError("out must have as many elements as x")
    _lib.check(_lib._launch(_lib.lib().igx_clamp, dev, _lib.tensor_ptr(x, torch.float32), C.c_float(lo), C.c_float(hi),
                            _lib.tensor_ptr(out, torch.float32), C.c_int64(x.numel())))


# ---- AntImitate / global quantile ---------------------------------------------------------
def imitate_args(motion, target_pose_idx, pose_diff, workspace, quantile, distance_floor, death_cost, n_total=None):
    """igx_imitate_args over device tensors (motion [F,8] f64, target_pose_idx [n] i64, pose_diff [n] f64,
    workspace >= IGX_QUANTILE_WORKSPACE_BYTES)."""
    if workspace.numel() * workspace.element_size() < _abi.IGX_QUANTILE_WORKSPACE_BYTES:
        raise ValueError("workspace smaller than IGX_QUANTILE_WORKSPACE_BYTES")
    if motion.dim() != 2 or motion.shape[1] != 8:
        raise ValueError("motion must be [num_frames, 8]")
    n = target_pose_idx.shape[0]
    if pose_diff.shape[0] != n:
        raise ValueError("pose_diff and target_pose_idx must have one entry per environment")
    return _abi.ImitateArgs(motion=_lib.tensor_ptr(motion, torch.float64), num_frames=motion.shape[0],
                            target_pose_idx=_lib.tensor_ptr(target_pose_idx, torch.int64),
                            pose_diff=_lib.tensor_ptr(pose_diff, torch.float64), workspace=_lib.tensor_ptr(workspace),
                            quantile=quantile, distance_floor=distance_floor, death_cost=death_cost,
                            n_total=n if n_total is None else n_total)


def _imitate_call(symbol, params, tensors, im, n_envs, cache=None):
    """As _lib.call_step, with the imitation arguments as a third struct (`cache`: see there)."""
    if cache is not None:
        ids = tuple(map(id, tensors.values()))
        ent = cache.get(symbol)
        if ent is not None and ent[0] == ids and ent[1] == n_envs and _lib._pointers(ent[6]) == ent[7]:
            _lib.check(_lib._launch(ent[2], ent[4], C.byref(params), ent[3], C.byref(im), C.c_int64(n_envs)))
            return
    _lib.check_sizes(_abi.LocoBuffers, params, tensors, n_envs, symbol)
    b, dev = _lib.make_buffers(_abi.LocoBuffers, tensors)
    fn = getattr(_lib.lib(), symbol)
    if cache is not None:
        held = _lib._held(tensors.values())
        cache[symbol] = (ids, n_envs, fn, C.byref(b), dev, b, held, _lib._pointers(held), _lib._storages(held))
    _lib.check(_lib._launch(fn, dev, C.byref(params), C.byref(b), C.byref(im), C.c_int64(n_envs)))


def ant_imitate_step(params, tensors, im, n_envs, cache=None):
    _imitate_call("igx_ant_imitate_step", params, tensors, im, n_envs, cache)


def ant_imitate_observe(params, tensors, im, n_envs, cache=None):
    _imitate_call("igx_ant_imitate_observe", params, tensors, im, n_envs, cache)


def ant_imitate_finish(params, tensors, im, n_envs, cache=None):
    _imitate_call("igx_ant_imitate_finish", params, tensors, im, n_envs, cache)


def quantile_f64(x, q, workspace=None):
    """torch.quantile(x, q) of a 1-D fp64 device tensor: exact radix select (NaN if any element is NaN);
    returns a 0-dim tensor."""
    dev = x.device
    if workspace is None:
        workspace = torch.empty(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=dev)
    out = torch.empty((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_quantile_f64(_lib.tensor_ptr(x, torch.float64), C.c_int64(x.numel()), C.c_double(q),
                                         _lib.tensor_ptr(out), _lib.tensor_ptr(workspace), _lib.current_stream(dev))
    _lib.check(rc)
    return out


def quantile_f64_windowed(x, q, workspace):
    """The same value as quantile_f64 through igx_quantile_f64_windowed: fewer launches when the same workspace is
    used call after call on data that drifts (the window the select keeps there follows the value); returns a 0-dim
    tensor."""
    if workspace.numel() * workspace.element_size() < _abi.IGX_QUANTILE_WORKSPACE_BYTES:
        raise ValueError("workspace smaller than IGX_QUANTILE_WORKSPACE_BYTES")
    dev = x.device
    out = torch.empty((), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_quantile_f64_windowed(_lib.tensor_ptr(x, torch.float64), C.c_int64(x.numel()), C.c_double(q),
                                                  _lib.tensor_ptr(out), _lib.tensor_ptr(workspace), _lib.current_stream(dev))
    _lib.check(rc)
    return out


def quantile_sharded(x, q, n_total, workspace, all_reduce):
    """The same select over the shards of a multi-GPU job: `all_reduce(hist)` must sum the int64 histogram
    tensor over the ranks in place (torch.distributed.all_reduce).  The result is left in the workspace
    state (read by ant_imitate_finish) and returned as a 0-dim tensor."""
    dev = workspace.device
    L = _lib.lib()
    ws = _lib.tensor_ptr(workspace)
    words = workspace.view(torch.int64)
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        _lib.check(L.igx_quantile_begin(ws, C.c_double(q), C.c_int64(n_total), st))
        for p in range(_abi.IGX_QUANTILE_PASSES):
            _lib.check(L.igx_quantile_histogram(_lib.tensor_ptr(x, torch.float64), C.c_int64(x.numel()), ws, C.c_int32(p), st))
            off = _abi.IGX_QUANTILE_HIST_OFFSET(p) // 8
            all_reduce(words[off:off + 2 * 2048])
        _lib.check(L.igx_quantile_finish(ws, None, st))
    return workspace.view(torch.float64)[5].clone()


_STATUS_WORD = 19  # int32 index of WindowState::status in the workspace (quantile.hip)


class WindowQuantile:
    """torch.quantile over the shards of a multi-GPU job in two collectives: one all-reduce of a 16 384-bin
    histogram over a window of the key space that all ranks share, one all-gather of the keys in the bin
    that holds the wanted rank (igx_quantile_window_*).  The window follows the value from call to call;
    a first call, or a value that left the window, costs further histogram rounds, and data the window
    cannot resolve (IGX_QW_GIVE_UP) goes to the six-pass select of `quantile_sharded`.  The bits are those
    of `quantile_f64` over the concatenated shards.  One object per (tensor stream, quantile): the window
    is state."""

    def __init__(self, device, capacity=_abi.IGX_QUANTILE_WINDOW_MAX_CANDIDATES):
        if not 1 <= capacity <= _abi.IGX_QUANTILE_WINDOW_MAX_CANDIDATES:
            raise ValueError("capacity outside [1, %d]" % _abi.IGX_QUANTILE_WINDOW_MAX_CANDIDATES)
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.workspace = self._alloc(_abi.IGX_QUANTILE_WINDOW_WORKSPACE_BYTES // 4, torch.int32)
        self.decision = self._alloc(4, torch.int64)
        self.slots = self._alloc(_abi.IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2, torch.int64)
        self.collectives = 0   # of the last run()
        self.rounds = 0
        self.fallbacks = 0     # runs that ended in the six-pass select

    def _alloc(self, n, dtype):
        return torch.zeros(n, dtype=dtype, device=self.device)

    # -- the five kernels (tests replace these with the CPU restatement) --
    def _call(self, name, *args):
        with torch.cuda.device(self.device):
            _lib.check(getattr(_lib.lib(), name)(*args, _lib.current_stream(self.device)))

    def begin(self, q, n_total, reset):
        self._call("igx_quantile_window_begin", _lib.tensor_ptr(self.workspace), C.c_double(q), C.c_int64(n_total),
                   C.c_int32(1 if reset else 0))

    def histogram(self, x):
        self._call("igx_quantile_window_histogram", _lib.tensor_ptr(x, torch.float64), C.c_int64(x.numel()),
                   _lib.tensor_ptr(self.workspace))

    def locate(self):
        self._call("igx_quantile_window_locate", _lib.tensor_ptr(self.workspace), C.c_int32(self.capacity),
                   _lib.tensor_ptr(self.decision))

    def collect(self, x, num_slots):
        self._call("igx_quantile_window_collect", _lib.tensor_ptr(x, torch.float64), C.c_int64(x.numel()),
                   _lib.tensor_ptr(self.workspace), _lib.tensor_ptr(self.slots), C.c_int64(num_slots))

    def finish(self, gathered, num_ranks, num_slots, out):
        self._call("igx_quantile_window_finish", _lib.tensor_ptr(self.workspace), _lib.tensor_ptr(gathered, torch.int64),
                   C.c_int32(num_ranks), C.c_int64(num_slots), _lib.tensor_ptr(out, torch.float64) if out is not None else None)

    # -- the protocol --
    def hist_view(self):
        off = _abi.IGX_QUANTILE_WINDOW_HIST_OFFSET // 4
        return self.workspace[off:off + _abi.IGX_QUANTILE_WINDOW_HIST_WORDS]

    def result(self):
        """The value of the last run() (0-dim fp64 view of the workspace)."""
        return self.workspace.view(torch.float64)[6]

    def run(self, x, q, n_total, all_reduce, all_gather, out=None, reset=False):
        """`all_reduce(t)` sums an int32 tensor over the ranks in place; `all_gather(t)` returns the [world, len(t)]
        int64 tensor of every rank's `t` in rank order.  Returns True when the window protocol produced the value
        (in `out` when given, and in result()), False when the caller has to run the six-pass select."""
        if x.dtype != torch.float64 or not x.is_contiguous():
            raise ValueError("x must be a contiguous fp64 tensor")
        self.collectives = self.rounds = 0
        self.begin(q, n_total, reset)
        for _ in range(_abi.IGX_QUANTILE_WINDOW_MAX_ROUNDS + 1):
            self.histogram(x)
            all_reduce(self.hist_view())
            self.collectives += 1
            self.rounds += 1
            self.locate()
            status, count = (int(v) for v in self.decision[:2].tolist())  # synchronises
            if status == _abi.IGX_QW_REFINED:
                continue
            if status in (_abi.IGX_QW_COLLECT, _abi.IGX_QW_VALUE):
                num_slots = count + 2
                self.collect(x, num_slots)
                gathered = all_gather(self.slots[:num_slots]).contiguous()
                self.collectives += 1
                self.finish(gathered, gathered.shape[0], num_slots, out)
                if _lib.DEBUG_CHECKS:  # synchronises: the finish kernel marks a gather that does not add up
                    state = int(self.workspace[_STATUS_WORD])
                    if state not in (_abi.IGX_QW_COLLECT, _abi.IGX_QW_VALUE):
                        raise RuntimeError("window quantile: the gathered slots do not hold the located bin "
                                           "(status %d); did every rank gather the same tensor?" % state)
                return True
            break
        self.fallbacks += 1
        return False
