"""Drop-in counterparts of the reference's utils/torch_jit_utils.py functions,
computed by the HIP kernels of libigx.so (no torch fallback).

Same names, argument order and result shapes as the reference
(IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py; line numbers at each
function).  Inputs are fp32 device tensors; quaternions are xyzw.
"""
import torch

from .. import _lib


def _rows(t, width):
    if t.shape[-1] != width:
        raise ValueError("expected last dimension %d, got shape %s" % (width, tuple(t.shape)))
    return t.reshape(-1, width).contiguous().float()


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
    xs = x.reshape(-1, d).contiguous().float()
    lo = lower.expand(d).contiguous().float()
    hi = upper.expand(d).contiguous().float()
    out = torch.empty_like(xs)
    _lib.call_affine(symbol, xs, lo, hi, out, xs.shape[0], d)
    return out.view(x.shape)


def scale(x, lower, upper):  # :234-236
    return _affine("igx_scale", x, lower, upper)


def unscale(x, lower, upper):  # :239-241
    return _affine("igx_unscale", x, lower, upper)


def tensor_clamp(t, min_t, max_t):  # :228-230
    return _affine("igx_tensor_clamp", t, min_t, max_t)


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):  # :37-38
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)
