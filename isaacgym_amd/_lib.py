"""Loader and thin call layer for libigx.so (the HIP kernels behind the C ABI).

There is no fallback: if the shared library is missing, or a tensor is not a
contiguous device tensor of the documented dtype, the call raises.
"""
import ctypes as C
import os

import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# IGX_LIB overrides the library path (used to A/B differently tuned builds)
LIB_PATH = os.environ.get("IGX_LIB") or os.path.join(_HERE, "libigx.so")
_lib = None

# every symbol include/igx.h declares: (name, restype, argtypes)
_P = C.c_void_p
_STEP_SIG = (C.c_int, [_P, _P, C.c_int64, _P])
_RESET_SIG = (C.c_int, [_P, _P, _P, C.c_int64, _P])
SYMBOLS = {
    "igx_version": (C.c_int, []),
    "igx_last_error_string": (C.c_char_p, []),
    "igx_cartpole_step": _STEP_SIG,
    "igx_cartpole_reset": _RESET_SIG,
    "igx_ant_reset": _RESET_SIG,
    "igx_humanoid_reset": _RESET_SIG,
    "igx_ant_step": _STEP_SIG,
    "igx_humanoid_step": _STEP_SIG,
    "igx_allegro_pre_physics": _STEP_SIG,
    "igx_allegro_post_physics": _STEP_SIG,
    "igx_allegro_post_physics_static": _STEP_SIG,
    "igx_allegro_static_prepare": (C.c_int, []),
    "igx_allegro_post_physics_shard": _STEP_SIG,
    "igx_allegro_finalize_shards": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "igx_allegro_finalize_shards_queued": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "igx_shadow_finalize_shards_queued": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "igx_shadow_pre_physics": _STEP_SIG,
    "igx_shadow_post_physics": _STEP_SIG,
    "igx_shadow_post_physics_shard": _STEP_SIG,
    "igx_shadow_finalize_shards": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
}
SYMBOLS["igx_vec_task_tail_step"] = (C.c_int, [_P, C.c_int64, _P])
SYMBOLS["igx_clamp"] = (C.c_int, [_P, C.c_float, C.c_float, _P, C.c_int64, _P])
SYMBOLS["igx_noise_apply"] = (C.c_int, [_P, _P, _P, C.c_int64, _P])
SYMBOLS["igx_action_stage_step"] = (C.c_int, [_P, C.c_int64, _P])
SYMBOLS.update({
    "igx_humanoid_imitate_step": _STEP_SIG,
    "igx_ant_imitate_step": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "igx_ant_imitate_observe": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "igx_ant_imitate_finish": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "igx_quantile_begin": (C.c_int, [_P, C.c_double, C.c_int64, _P]),
    "igx_quantile_histogram": (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P]),
    "igx_quantile_finish": (C.c_int, [_P, _P, _P]),
    "igx_quantile_f64": (C.c_int, [_P, C.c_int64, C.c_double, _P, _P, _P]),
    "igx_quantile_f64_windowed": (C.c_int, [_P, C.c_int64, C.c_double, _P, _P, _P]),
    "igx_quantile_window_begin": (C.c_int, [_P, C.c_double, C.c_int64, C.c_int32, _P]),
    "igx_quantile_window_histogram": (C.c_int, [_P, C.c_int64, _P, _P]),
    "igx_quantile_window_locate": (C.c_int, [_P, C.c_int32, _P, _P]),
    "igx_quantile_window_collect": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P]),
    "igx_quantile_window_finish": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P]),
})
SYMBOLS["igx_episode_stats_step"] = (C.c_int, [_P, C.c_int64, _P])
SYMBOLS["igx_episode_window_step"] = (C.c_int, [_P, C.c_int64, _P])
SYMBOLS["igx_hand_random_forces"] = (C.c_int, [_P, C.c_int64, _P])
# trajectory recorder
SYMBOLS.update({
    "igx_recorder_create": (C.c_int, [C.POINTER(_P), C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int64]),
    "igx_recorder_capture": (C.c_int, [_P, _P, _P]),
    "igx_recorder_sync": (C.c_int, [_P]),
    "igx_recorder_steps": (C.c_int64, [_P]),
    "igx_recorder_host_data": (_P, [_P]),
    "igx_recorder_destroy": (C.c_int, [_P]),
})
# AMP motion library / HumanoidAMP
SYMBOLS.update({
    "igx_motion_state": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "igx_amp_observations": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, C.c_int64, _P]),
    "igx_motion_amp_obs": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, C.c_int64, _P]),
    "igx_humanoid_amp_step": _STEP_SIG,
    "igx_motion_sample_flagged": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P, _P]),
    "igx_motion_sample": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "igx_motion_amp_obs_steps": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_int32, _P, _P]),
    "igx_motion_fetch_amp_obs": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_double, C.c_int32, _P, _P, _P, _P]),
    "igx_motion_amp_obs_history": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_int32, _P, _P]),
    "igx_humanoid_amp_reset": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
})
# batched torch_jit_utils math: (inputs..., out, n[, d], stream)
_I64 = C.c_int64
for _name, _nin in (("igx_quat_mul", 2), ("igx_quat_conjugate", 1), ("igx_quat_rotate", 2),
                    ("igx_quat_rotate_inverse", 2), ("igx_slerp", 3), ("igx_quat_from_angle_axis", 2),
                    ("igx_normalize_angle", 1), ("igx_quat_to_exp_map", 1), ("igx_exp_map_to_quat", 1),
                    ("igx_quat_to_tan_norm", 1), ("igx_calc_heading_quat_inv", 1)):
    SYMBOLS[_name] = (C.c_int, [_P] * (_nin + 1) + [_I64, _P])
for _name in ("igx_scale", "igx_unscale", "igx_tensor_clamp", "igx_scale_transform", "igx_unscale_transform",
              "igx_saturate"):
    SYMBOLS[_name] = (C.c_int, [_P, _P, _P, _P, _I64, C.c_int32, _P])
# (name, number of input arrays + output arrays): all pointers, then n and the stream
for _name, _narr in (("igx_quat_unit", 2), ("igx_quat_apply", 3), ("igx_tf_inverse", 4), ("igx_tf_apply", 4),
                     ("igx_tf_combine", 6), ("igx_get_euler_xyz", 4), ("igx_quat_from_euler_xyz", 4),
                     ("igx_euler_xyz_to_exp_map", 4), ("igx_compute_rot", 11), ("igx_quat_diff_rad", 3),
                     ("igx_local_to_world_space", 3), ("igx_quat_to_angle_axis", 3), ("igx_angle_axis_to_exp_map", 3),
                     ("igx_exp_map_to_angle_axis", 3), ("igx_calc_heading", 2), ("igx_calc_heading_quat", 2)):
    SYMBOLS[_name] = (C.c_int, [_P] * _narr + [_I64, _P])
SYMBOLS["igx_normalize"] = (C.c_int, [_P, _P, _I64, C.c_int32, _P])
SYMBOLS["igx_copysign"] = (C.c_int, [C.c_float, _P, _P, _I64, _P])
SYMBOLS["igx_quat_axis"] = (C.c_int, [_P, C.c_int32, _P, _I64, _P])
SYMBOLS["igx_extract_positions_from_rigid_body_states"] = (C.c_int, [_P, _I64, C.c_int32, _P, _I64, _P])
SYMBOLS["igx_torch_rand_float"] = (C.c_int, [C.c_double, C.c_double, _P, C.c_uint64, C.c_uint64, _P, _I64, _P])
SYMBOLS["igx_compute_heading_and_up"] = (C.c_int, [_P] * 5 + [C.c_int32] + [_P] * 5 + [_I64, _P])


class IgxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libigx error %d: %s" % (code, text))
        self.code = code


def lib():
    """Load libigx.so once; raise ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C isaacgym_amd/csrc` (or __graft_entry__.build()); "
                "isaacgym_amd has no CPU fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise IgxError(rc, lib().igx_last_error_string().decode("utf-8", "replace"))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device):
    """torch's current stream on `device` as the raw hipStream_t the C ABI takes."""
    if _raw_stream is not None:
        index = device.index if isinstance(device, torch.device) else torch.device(device).index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if index is None else index))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def tensor_ptr(t, dtype=None):
    """Raw device address of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor, got %r" % type(t))
    if not t.is_cuda:
        raise ValueError("libigx works on device tensors only (got a %s tensor)" % t.device)
    if not t.is_contiguous():
        raise ValueError("libigx needs contiguous tensors; pass the base tensor, not a strided view")
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


_I64_FIELDS = {"reset", "progress", "randomize", "reset_goal", "terminate", "motion_ids", "frame_idx0",
               "frame_idx1", "object_indices", "goal_object_indices", "env_ids", "done"}
_RAW_FIELDS = {"workspace", "shard_record"}  # opaque device scratch: any dtype
_BOOL_FIELDS = {"time_outs"}
_F64_FIELDS = {"motion_times", "motion_frame"}


def _wanted_dtype(name):
    if name in _RAW_FIELDS:
        return None
    return (torch.int64 if name in _I64_FIELDS else torch.float64 if name in _F64_FIELDS else
            torch.bool if name in _BOOL_FIELDS else torch.float32)


_FIELD_SPECS = {}  # struct class -> {field name: wanted dtype (None = any)}


def make_buffers(cls, tensors):
    """Build a *Buffers struct from name->tensor; checks device/dtype/contiguity
    and that every tensor lives on the same device."""
    spec = _FIELD_SPECS.get(cls)
    if spec is None:
        spec = _FIELD_SPECS[cls] = {n: _wanted_dtype(n) for n, _ in cls._fields_}
    b = cls()
    dev = None
    for k, t in tensors.items():
        if t is None:
            continue
        try:
            want = spec[k]
        except KeyError:
            raise KeyError("%s has no buffer named %r" % (cls.__name__, k)) from None
        # the checks of tensor_ptr, inlined: this loop runs for every buffer of every step
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and (want is None or t.dtype is want)):
            tensor_ptr(t, want)  # raises the matching error
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("buffer %r is on %s, others on %s" % (k, t.device, dev))
        setattr(b, k, t.data_ptr())
    for k in tensors:
        if k not in spec:
            raise KeyError("%s has no buffer named %r" % (cls.__name__, k))
    return b, dev


# IGX_DEBUG_CHECKS=1: also the checks that have to read device memory (and therefore synchronise):
# index tensors against the number of rows they address
DEBUG_CHECKS = os.environ.get("IGX_DEBUG_CHECKS", "0") not in ("", "0")


def check_sizes(bufcls, params, tensors, n, who):
    """Every given buffer must hold what a call on n environments reads or writes (a short buffer
    would be an out-of-bounds device access, not an error code)."""
    if n < 0:
        raise ValueError("%s: n_envs %d is negative" % (who, n))
    need = _abi.required_elements(bufcls, params, n)
    if need is None:
        return
    for k, t in tensors.items():
        if t is None:
            continue
        want = need.get(k)
        if want is None:
            continue
        count, is_bytes = want
        have = t.numel() * t.element_size() if is_bytes else t.numel()
        if have < count:
            raise ValueError("%s: buffer %r holds %d %s, a call on %d environments needs %d" %
                             (who, k, have, "bytes" if is_bytes else "elements", n, count))
    if DEBUG_CHECKS and n > 0:
        rows = tensors.get("root_state_tensor")
        for k in ("object_indices", "goal_object_indices"):
            idx = tensors.get(k)
            if idx is not None and rows is not None and idx.numel():
                lo, hi = int(idx[:n].min()), int(idx[:n].max())
                if lo < 0 or hi >= rows.numel() // 13:
                    raise ValueError("%s: %s addresses rows %d..%d of a %d-row root_state_tensor" %
                                     (who, k, lo, hi, rows.numel() // 13))


def check_env_ids(bufcls, params, tensors, env_ids, who):
    """reset calls: the buffers are addressed by environment id.  The id list is device memory, so its
    range is only checked with IGX_DEBUG_CHECKS=1 (it synchronises); the buffers must at least be
    mutually consistent (all hold the same number of environments)."""
    widths = _abi.BUFFER_WIDTHS.get(bufcls)
    if widths is None:
        return
    envs = None
    for k, (kind, w) in widths(params).items():
        t = tensors.get(k)
        if t is None or kind != "env" or w == 0:
            continue
        have = t.numel() // w
        envs = have if envs is None else min(envs, have)
    if envs is None:
        return
    if DEBUG_CHECKS and env_ids.numel():
        hi = int(env_ids.max())
        if hi >= envs:
            raise ValueError("%s: env_ids reach %d, the smallest buffer holds %d environments" % (who, hi, envs))
    if env_ids.numel() > 0 and envs == 0:
        raise ValueError("%s: env_ids given for buffers that hold no environment" % who)


def _launch(fn, dev, *args):
    """fn(*args, stream) with `dev` current (the context switch is skipped when it already is)."""
    index = dev.index
    if index is None or index == torch.cuda.current_device():
        return fn(*args, current_stream(dev))
    with torch.cuda.device(dev):
        return fn(*args, current_stream(dev))


_TENSOR_DATA_PTR = torch.Tensor.data_ptr  # the unbound slot: map() calls it without a Python frame per tensor


def _held(tensors):
    """The tensors among `tensors` (absent optional buffers are None), as a cache entry keeps them."""
    return [t for t in tensors if t is not None]


def _pointers(held):
    """Where each tensor of a cache entry points now (the per-call re-validation of the cached structs)."""
    return list(map(_TENSOR_DATA_PTR, held))


def _storages(tensors):
    """The storages behind `tensors`, to be kept with a cache entry: while they are alive the allocator cannot hand
    their addresses to anything else, so a tensor re-pointed to NEW memory always shows a new data_ptr()."""
    return [t.untyped_storage() for t in tensors if t is not None]


def call_step(symbol, params, bufcls, tensors, n, cache=None):
    """One step entry point.  `cache` (a dict owned by the caller, one per task) keeps the validated buffer
    struct between calls: as long as the same tensor objects come back in `tensors` and n is unchanged, a
    call costs one tuple comparison instead of ~20 checks and attribute stores.  The size-determining
    parameters (observation layout, dof counts) must not change under a cache - they are what the task's
    buffers were allocated from."""
    if cache is not None:
        ids = tuple(map(id, tensors.values()))
        ent = cache.get(symbol)
        # same tensor objects, same n, and every one of them still points where it did when the struct was
        # validated (`t.data = ...`, set_() and resize_() keep a tensor's identity but move its memory)
        if ent is not None and ent[0] == ids and ent[1] == n and _pointers(ent[6]) == ent[7]:
            check(_launch(ent[2], ent[4], C.byref(params), ent[3], C.c_int64(n)))
            return
    check_sizes(bufcls, params, tensors, n, symbol)
    b, dev = make_buffers(bufcls, tensors)
    if dev is None:
        raise ValueError("no device buffers given")
    fn = getattr(lib(), symbol)
    if cache is not None:  # the entry keeps the tensors alive, so their ids cannot be reused while it exists
        held = _held(tensors.values())
        cache[symbol] = (ids, n, fn, C.byref(b), dev, b, held, _pointers(held), _storages(held))
    check(_launch(fn, dev, C.byref(params), C.byref(b), C.c_int64(n)))


def call_reset(symbol, params, bufcls, tensors, env_ids):
    check_env_ids(bufcls, params, tensors, env_ids, symbol)
    b, dev = make_buffers(bufcls, tensors)
    if dev is None:
        raise ValueError("no device buffers given")
    if env_ids.device != dev:
        raise ValueError("env_ids is on %s, buffers on %s" % (env_ids.device, dev))
    with torch.cuda.device(dev):
        rc = getattr(lib(), symbol)(C.byref(params), C.byref(b), tensor_ptr(env_ids, torch.int64),
                                    C.c_int64(env_ids.shape[0]), current_stream(dev))
    check(rc)


def call_rowwise(symbol, inputs, out, n):
    """inputs: contiguous fp32 device tensors; out: preallocated device tensor."""
    dev = out.device
    ptrs = [tensor_ptr(t, torch.float32) for t in inputs] + [tensor_ptr(out, torch.float32)]
    for t in inputs:
        if t.device != dev:
            raise ValueError("all tensors must live on %s" % dev)
    with torch.cuda.device(dev):
        rc = getattr(lib(), symbol)(*ptrs, C.c_int64(n), current_stream(dev))
    check(rc)


def call_arrays(symbol, inputs, outputs, n, before=(), between=()):
    """symbol(*before, inputs..., *between, outputs..., n, stream) on contiguous fp32 device tensors."""
    dev = outputs[0].device
    for t in list(inputs) + list(outputs):
        if t.device != dev:
            raise ValueError("all tensors must live on %s" % dev)
    args = (list(before) + [tensor_ptr(t, torch.float32) for t in inputs] + list(between) +
            [tensor_ptr(t, torch.float32) for t in outputs])
    with torch.cuda.device(dev):
        rc = getattr(lib(), symbol)(*args, C.c_int64(n), current_stream(dev))
    check(rc)


def call_affine(symbol, x, lower, upper, out, n, d):
    dev = out.device
    ptrs = [tensor_ptr(t, torch.float32) for t in (x, lower, upper, out)]
    with torch.cuda.device(dev):
        rc = getattr(lib(), symbol)(*ptrs, C.c_int64(n), C.c_int32(d), current_stream(dev))
    check(rc)
