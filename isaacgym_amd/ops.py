"""Low-level Python entry points: one function per C-ABI call of libigx.so.

`params` is the matching isaacgym_amd._abi.*Params struct, `tensors` maps the
field names of the *Buffers struct to contiguous device tensors.
"""
import ctypes as C

import torch

from . import _abi, _lib


def cartpole_step(params, tensors, n_envs):
    _lib.call_step("igx_cartpole_step", params, _abi.CartpoleBuffers, tensors, n_envs)


def ant_step(params, tensors, n_envs):
    _lib.call_step("igx_ant_step", params, _abi.LocoBuffers, tensors, n_envs)


def humanoid_step(params, tensors, n_envs):
    _lib.call_step("igx_humanoid_step", params, _abi.LocoBuffers, tensors, n_envs)


def allegro_pre_physics(params, tensors, n_envs):
    _lib.call_step("igx_allegro_pre_physics", params, _abi.AllegroBuffers, tensors, n_envs)


def allegro_post_physics(params, tensors, n_envs):
    _lib.call_step("igx_allegro_post_physics", params, _abi.AllegroBuffers, tensors, n_envs)


# ---- AMP motion library / HumanoidAMP ---------------------------------------
_TABLE_DTYPES = dict(motion_length=torch.float64, motion_dt=torch.float64, motion_num_frames=torch.int64,
                     frame_offset=torch.int64)


def motion_tables(tensors):
    """igx_motion_tables from name -> device tensor (the caller keeps the tensors alive)."""
    t = _abi.MotionTables()
    t.num_motions = int(tensors["motion_length"].shape[0])
    for name, ctype in _abi.MotionTables._fields_:
        if ctype is _abi.ptr:
            setattr(t, name, _lib.tensor_ptr(tensors[name], _TABLE_DTYPES.get(name, torch.float32)))
    return t


def motion_state(tables, motion_ids, motion_times, out):
    """out: name -> preallocated device tensor for any subset of igx_motion_state_out."""
    o, _ = _lib.make_buffers(_abi.MotionStateOut, out)
    dev = motion_ids.device
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_state(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                         _lib.tensor_ptr(motion_times, torch.float64),
                                         C.c_int64(motion_ids.shape[0]), C.byref(o), _lib.current_stream(dev))
    _lib.check(rc)


def motion_amp_obs(tables, motion_ids, motion_times, local_root_obs, out, out_rows=None, out_row_stride=None,
                   out_col_offset=0):
    dev = motion_ids.device
    stride = _abi.IGX_AMP_OBS if out_row_stride is None else out_row_stride
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_motion_amp_obs(C.byref(tables), _lib.tensor_ptr(motion_ids, torch.int64),
                                           _lib.tensor_ptr(motion_times, torch.float64),
                                           C.c_int64(motion_ids.shape[0]), C.c_int32(int(bool(local_root_obs))),
                                           _lib.tensor_ptr(out, torch.float32),
                                           _lib.tensor_ptr(out_rows, torch.int64), C.c_int64(stride),
                                           C.c_int64(out_col_offset), _lib.current_stream(dev))
    _lib.check(rc)


def amp_observations(root_states, dof_pos, dof_vel, key_body_pos, local_root_obs, obs):
    dev = obs.device
    f = torch.float32
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_amp_observations(_lib.tensor_ptr(root_states, f), _lib.tensor_ptr(dof_pos, f),
                                             _lib.tensor_ptr(dof_vel, f), _lib.tensor_ptr(key_body_pos, f),
                                             C.c_int32(int(bool(local_root_obs))), _lib.tensor_ptr(obs, f),
                                             C.c_int64(obs.shape[0]), _lib.current_stream(dev))
    _lib.check(rc)


def humanoid_amp_step(params, tensors, n_envs):
    _lib.call_step("igx_humanoid_amp_step", params, _abi.AmpBuffers, tensors, n_envs)


def humanoid_amp_reset(params, tensors, env_ids, state):
    """state: root_pos / root_rot / dof_pos / root_vel / root_ang_vel / dof_vel rows for env_ids."""
    b, dev = _lib.make_buffers(_abi.AmpBuffers, tensors)
    st, _ = _lib.make_buffers(_abi.MotionStateOut, state)
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_humanoid_amp_reset(C.byref(params), C.byref(b), _lib.tensor_ptr(env_ids, torch.int64),
                                               C.c_int64(env_ids.shape[0]), C.byref(st), _lib.current_stream(dev))
    _lib.check(rc)
