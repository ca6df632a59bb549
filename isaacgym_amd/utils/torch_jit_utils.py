"""Drop-in counterparts of the reference's utils/torch_jit_utils.py functions,
computed by the HIP kernels of libigx.so (no torch fallback).

Same names, argument order and result shapes as the reference
(IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py; line numbers at each
function).  Inputs are fp32 device tensors; quaternions are xyzw.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def _f32(t):
    """The kernels compute in fp32; a float64 input would come back with less precision than the reference's
    result for it, so it is refused instead of being narrowed silently."""
    if t.dtype == torch.float64:
        raise TypeError("float64 input: the kernels behind these functions compute in float32")
    return t.contiguous().float()


def _rows(t, width):
    if t.shape[-1] != width:
        raise ValueError("expected last dimension %d, got shape %s" % (width, tuple(t.shape)))
    return _f32(t.reshape(-1, width))


def _run(symbol, ins, widths, out_width, out_shape):
    rows = [_rows(t, w) for t, w in zip(ins, widths)]
    n = rows[0].shape[0]
    for r in rows:
        if r.shape[0] != n:
            raise ValueError("batch sizes differ: %s" % [tuple(r.shape) for r in rows])
    out = torch.empty((n, out_width), dtype=torch.float32, device=rows[0].device)
    _lib.call_rowwise(symbol, rows, out, n)
    return out.view(out_shape)


def quat_mul(a, b):  # :41-62
    assert a.shape == b.shape
    return _run("igx_quat_mul", (a, b), (4, 4), 4, a.shape)


def quat_conjugate(a):  # :106-110
    return _run("igx_quat_conjugate", (a,), (4,), 4, a.shape)


def quat_rotate(q, v):  # :80-90
    return _run("igx_quat_rotate", (q, v), (4, 3), 3, v.shape)


def my_quat_rotate(q, v):  # :411-421 (same arithmetic as quat_rotate)
    return quat_rotate(q, v)


def quat_rotate_inverse(q, v):  # :93-103
    return _run("igx_quat_rotate_inverse", (q, v), (4, 3), 3, v.shape)


def slerp(q0, q1, t):  # :595-628; t broadcasts against q0[..., :1]
    t = t.expand(q0.shape[:-1] + (1,))
    return _run("igx_slerp", (q0, q1, t), (4, 4, 1), 4, q0.shape)


def quat_from_angle_axis(angle, axis):  # :118-123
    return _run("igx_quat_from_angle_axis", (angle.unsqueeze(-1), axis), (1, 3), 4, axis.shape[:-1] + (4,))


def normalize_angle(x):  # :126-128
    return _run("igx_normalize_angle", (x.unsqueeze(-1),), (1,), 1, x.shape)


def quat_to_exp_map(q):  # :454-461
    return _run("igx_quat_to_exp_map", (q,), (4,), 3, q.shape[:-1] + (3,))


def exp_map_to_quat(exp_map):  # :589-593
    return _run("igx_exp_map_to_quat", (exp_map,), (3,), 4, exp_map.shape[:-1] + (4,))


def quat_to_tan_norm(q):  # :548-561
    return _run("igx_quat_to_tan_norm", (q,), (4,), 6, q.shape[:-1] + (6,))


def calc_heading_quat_inv(q):  # :656-667
    return _run("igx_calc_heading_quat_inv", (q,), (4,), 4, q.shape)


def _affine(symbol, x, lower, upper):
    d = x.shape[-1]
    if max(lower.dim(), upper.dim()) > 1 and x.numel() < 2 ** 31:
        # bounds that vary along the leading dimensions too (the reference's expressions broadcast them):
        # the whole array as one row, each element with a bound of its own
        lower, upper, d = lower.expand(x.shape).reshape(-1), upper.expand(x.shape).reshape(-1), max(x.numel(), 1)
    xs = _f32(x.reshape(-1, d))
    lo = _f32(lower.expand(d))
    hi = _f32(upper.expand(d))
    out = torch.empty_like(xs)
    _lib.call_affine(symbol, xs, lo, hi, out, xs.shape[0], d)
    return out.view(x.shape)


def scale(x, lower, upper):  # :234-236
    return _affine("igx_scale", x, lower, upper)


def unscale(x, lower, upper):  # :239-241
    return _affine("igx_unscale", x, lower, upper)


def tensor_clamp(t, min_t, max_t):  # :228-230
    return _affine("igx_tensor_clamp", t, min_t, max_t)


def _run_multi(symbol, ins, widths, out_widths, out_shapes, before=(), between=()):
    """Several input arrays -> several output arrays, one launch."""
    rows = [_rows(t, w) for t, w in zip(ins, widths)]
    n = rows[0].shape[0]
    for r in rows:
        if r.shape[0] != n:
            raise ValueError("batch sizes differ: %s" % [tuple(r.shape) for r in rows])
    dev = rows[0].device
    outs = [torch.empty((n, w), dtype=torch.float32, device=dev) for w in out_widths]
    _lib.call_arrays(symbol, rows, outs, n, before=before, between=between)
    return tuple(o.view(sh) for o, sh in zip(outs, out_shapes))


def normalize(x, eps: float = 1e-9):  # :65-67 (rows of 3 or 4 floats; eps is the reference's default)
    if eps != 1e-9:
        raise ValueError("the kernel implements the reference's default eps = 1e-9")
    d = x.shape[-1]
    xs = _rows(x, d)
    out = torch.empty_like(xs)
    dev = xs.device
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_normalize(_lib.tensor_ptr(xs, torch.float32), _lib.tensor_ptr(out, torch.float32),
                                      C.c_int64(xs.shape[0]), C.c_int32(d), _lib.current_stream(dev))
    _lib.check(rc)
    return out.view(x.shape)


def quat_unit(a):  # :113-115
    return _run_multi("igx_quat_unit", (a,), (4,), (4,), (a.shape,))[0]


def quat_apply(a, b):  # :70-77
    return _run_multi("igx_quat_apply", (a, b), (4, 3), (3,), (b.shape,))[0]


def tf_inverse(q, t):  # :131-134
    return _run_multi("igx_tf_inverse", (q, t), (4, 3), (4, 3), (q.shape, t.shape))


def tf_apply(q, t, v):  # :137-139
    return _run_multi("igx_tf_apply", (q, t, v), (4, 3, 3), (3,), (v.shape,))[0]


def tf_vector(q, v):  # :142-144
    return quat_apply(q, v)


def tf_combine(q1, t1, q2, t2):  # :147-149
    return _run_multi("igx_tf_combine", (q1, t1, q2, t2), (4, 3, 4, 3), (4, 3), (q1.shape, t1.shape))


def get_basis_vector(q, v):  # :152-154
    return quat_rotate(q, v)


def get_axis_params(value, axis_idx, x_value=0., dtype=float, n_dims=3):  # :157-165 (host-side list, as in the reference)
    zs = np.zeros((n_dims,))
    assert axis_idx < n_dims, "the axis dim should be within the vector dimensions"
    zs[axis_idx] = 1.
    params = np.where(zs == 1., value, zs)
    params[0] = x_value
    return list(params.astype(dtype))


def copysign(a: float, b):  # :168-172
    return _run_multi("igx_copysign", (b.unsqueeze(-1),), (1,), (1,), (b.shape,), before=(C.c_float(a),))[0]


def get_euler_xyz(q):  # :175-195
    sh = q.shape[:-1]
    return _run_multi("igx_get_euler_xyz", (q,), (4,), (1, 1, 1), (sh, sh, sh))


def quat_from_euler_xyz(roll, pitch, yaw):  # :198-212
    return _run_multi("igx_quat_from_euler_xyz", (roll.unsqueeze(-1), pitch.unsqueeze(-1), yaw.unsqueeze(-1)), (1, 1, 1),
                      (4,), (roll.shape + (4,),))[0]


def euler_xyz_to_exp_map(roll, pitch, yaw):  # :563-568
    return _run_multi("igx_euler_xyz_to_exp_map", (roll.unsqueeze(-1), pitch.unsqueeze(-1), yaw.unsqueeze(-1)), (1, 1, 1),
                      (3,), (roll.shape + (3,),))[0]


def torch_rand_float(lower, upper, shape, device, uniforms=None, seed=0, rng_offset=0):  # :215-218
    """(upper - lower) * u + lower.  The draws come from `uniforms` (what torch.rand(*shape) returned, to reproduce
    a torch sequence) or, without it, from the library's counter-based generator keyed by (seed, rng_offset)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    u = None
    if uniforms is not None:
        if tuple(uniforms.shape) != tuple(shape):
            raise ValueError("uniforms must have the requested shape")
        u = uniforms.contiguous().float()
    dev = out.device
    with torch.cuda.device(dev):
        rc = _lib.lib().igx_torch_rand_float(C.c_double(lower), C.c_double(upper), _lib.tensor_ptr(u), C.c_uint64(seed),
                                             C.c_uint64(rng_offset), _lib.tensor_ptr(out), C.c_int64(out.numel()),
                                             _lib.current_stream(dev))
    _lib.check(rc)
    return out


def compute_heading_and_up(torso_rotation, inv_start_rot, to_target, vec0, vec1, up_idx: int):  # :248-263
    n = torso_rotation.shape[0]
    return _run_multi("igx_compute_heading_and_up", (torso_rotation, inv_start_rot, to_target, vec0, vec1),
                      (4, 4, 3, 3, 3), (4, 1, 1, 3, 3), ((n, 4), (n,), (n,), (n, 3), (n, 3)),
                      between=(C.c_int32(up_idx),))


def compute_rot(torso_quat, velocity, ang_velocity, targets, torso_positions):  # :266-277
    n = torso_quat.shape[0]
    return _run_multi("igx_compute_rot", (torso_quat, velocity, ang_velocity, targets, torso_positions), (4, 3, 3, 3, 3),
                      (3, 3, 1, 1, 1, 1), ((n, 3), (n, 3), (n,), (n,), (n,), (n,)))


def quat_axis(q, axis: int = 0):  # :280-285
    return _run_multi("igx_quat_axis", (q,), (4,), (3,), ((q.shape[0], 3),), between=(C.c_int32(axis),))[0]


def scale_transform(x, lower, upper):  # :292-310
    return _affine("igx_scale_transform", x, lower, upper)


def unscale_transform(x, lower, upper):  # :313-331
    return _affine("igx_unscale_transform", x, lower, upper)


def saturate(x, lower, upper):  # :333-348
    return _affine("igx_saturate", x, lower, upper)


def quat_diff_rad(a, b):  # :354-373
    return _run_multi("igx_quat_diff_rad", (a, b), (4, 4), (1,), ((a.shape[0],),))[0]


def local_to_world_space(pos_offset_local, pose_global):  # :376-395
    return _run_multi("igx_local_to_world_space", (pos_offset_local, pose_global), (3, 7), (3,),
                      (pos_offset_local.shape,))[0]


def quat_to_angle_axis(q):  # :423-444
    return _run_multi("igx_quat_to_angle_axis", (q,), (4,), (1, 3), (q.shape[:-1], q.shape[:-1] + (3,)))


def angle_axis_to_exp_map(angle, axis):  # :446-452
    return _run_multi("igx_angle_axis_to_exp_map", (angle.unsqueeze(-1), axis), (1, 3), (3,), (axis.shape,))[0]


def exp_map_to_angle_axis(exp_map):  # :570-587
    return _run_multi("igx_exp_map_to_angle_axis", (exp_map,), (3,), (1, 3), (exp_map.shape[:-1], exp_map.shape))


def calc_heading(q):  # :630-641
    return _run_multi("igx_calc_heading", (q,), (4,), (1,), (q.shape[:-1],))[0]


def calc_heading_quat(q):  # :643-654
    return _run_multi("igx_calc_heading_quat", (q,), (4,), (4,), (q.shape,))[0]


def extract_positions_from_rigid_body_states(tensor, num_bodies: int = 22):  # :670-681 (22 bodies there)
    """[n, >= 13 * num_bodies] flattened rigid-body states -> [n, 3 * num_bodies] positions."""
    if tensor.dim() != 2 or tensor.shape[1] < 13 * num_bodies:
        raise IndexError("expected [n, >= %d], got %s" % (13 * num_bodies, tuple(tensor.shape)))
    x = tensor.contiguous().float()
    n = x.shape[0]
    out = torch.empty((n, 3 * num_bodies), dtype=torch.float32, device=x.device)
    fn = _lib.lib().igx_extract_positions_from_rigid_body_states
    _lib.check(_lib._launch(fn, x.device, _lib.tensor_ptr(x), C.c_int64(x.shape[1]), C.c_int32(num_bodies),
                            _lib.tensor_ptr(out), C.c_int64(n)))
    return out


def unscale_np(x, lower, upper):  # :244-245 (numpy on the host, as in the reference)
    return (2.0 * x - upper - lower) / (upper - lower)


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):  # :37-38
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)
