"""Low-level Python entry points: one function per C-ABI call of libigx.so.

`params` is the matching isaacgym_amd._abi.*Params struct, `tensors` maps the
field names of the *Buffers struct to contiguous device tensors.
"""
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
