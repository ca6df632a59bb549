"""CPU checker's torch_jit_utils functions against golden vectors produced by
the reference's own functions (tests/golden/math.npz)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H
from tests.math_cases import AFFINE, MULTI, ROWWISE, check_outputs, well_conditioned_rows


@pytest.mark.parametrize("fixture", ["math", "math_nonfinite", "math_extreme"])
@pytest.mark.parametrize("name,ins,outk,exact,scale", ROWWISE)
def test_rowwise(name, ins, outk, exact, scale, fixture):
    g = H.load_golden(fixture)
    got = oracle.rowwise(name, *[g[k] for k in ins])
    ref = g[outk]
    assert got.shape == ref.shape
    if exact:
        assert H.bit_equal_fraction(got, ref) == 1.0
    elif name == "normalize_angle":
        H.assert_angle_close(got, ref, what=name, branch_cut="pi")  # atan2(sin, cos): +-pi are one answer
    else:
        rows = well_conditioned_rows(name, g)
        H.assert_close(got[rows], ref[rows], scale=scale, what=name)


@pytest.mark.parametrize("fixture", ["math", "math_nonfinite", "math_extreme"])
@pytest.mark.parametrize("name", AFFINE)
def test_affine(name, fixture):
    g = H.load_golden(fixture)
    got = oracle.affine(name, g["aff_x"], g["aff_lo"], g["aff_hi"])
    assert H.bit_equal_fraction(got, g[name]) == 1.0


class OracleTJ:
    """The CPU checker behind the reference's Python signatures (numpy in, numpy out)."""

    @staticmethod
    def _one(name, ins, w, **kw):
        return oracle.arrays(name, ins, (w,), **kw)[0]

    normalize = staticmethod(lambda x: oracle.normalize(x))
    quat_unit = staticmethod(lambda a: OracleTJ._one("quat_unit", [a], 4))
    quat_apply = staticmethod(lambda a, b: OracleTJ._one("quat_apply", [a, b], 3))
    tf_vector = quat_apply
    tf_inverse = staticmethod(lambda q, t: oracle.arrays("tf_inverse", [q, t], (4, 3)))
    tf_apply = staticmethod(lambda q, t, v: OracleTJ._one("tf_apply", [q, t, v], 3))
    tf_combine = staticmethod(lambda q1, t1, q2, t2: oracle.arrays("tf_combine", [q1, t1, q2, t2], (4, 3)))
    get_basis_vector = staticmethod(lambda q, v: oracle.rowwise("quat_rotate", q, v))
    copysign = staticmethod(lambda a, b: OracleTJ._one("copysign", [b], 1, before=(C.c_float(a),)))
    get_euler_xyz = staticmethod(lambda q: oracle.arrays("get_euler_xyz", [q], (1, 1, 1)))
    quat_from_euler_xyz = staticmethod(lambda r, p, y: OracleTJ._one("quat_from_euler_xyz", [r, p, y], 4))
    euler_xyz_to_exp_map = staticmethod(lambda r, p, y: OracleTJ._one("euler_xyz_to_exp_map", [r, p, y], 3))
    compute_heading_and_up = staticmethod(lambda rot, isr, tt, v0, v1, idx: oracle.arrays(
        "compute_heading_and_up", [rot, isr, tt, v0, v1], (4, 1, 1, 3, 3), between=(C.c_int32(idx),)))
    compute_rot = staticmethod(lambda tq, v, a, t, p: oracle.arrays("compute_rot", [tq, v, a, t, p], (3, 3, 1, 1, 1, 1)))
    quat_axis = staticmethod(lambda q, axis: OracleTJ._one("quat_axis", [q], 3, between=(C.c_int32(axis),)))
    scale_transform = staticmethod(lambda x, lo, hi: oracle.affine("scale_transform", x, lo, hi))
    unscale_transform = staticmethod(lambda x, lo, hi: oracle.affine("unscale_transform", x, lo, hi))
    saturate = staticmethod(lambda x, lo, hi: oracle.affine("saturate", x, lo, hi))
    quat_diff_rad = staticmethod(lambda a, b: OracleTJ._one("quat_diff_rad", [a, b], 1))
    local_to_world_space = staticmethod(lambda p, pose: OracleTJ._one("local_to_world_space", [p, pose], 3))
    quat_to_angle_axis = staticmethod(lambda q: oracle.arrays("quat_to_angle_axis", [q], (1, 3)))
    angle_axis_to_exp_map = staticmethod(lambda a, ax: OracleTJ._one("angle_axis_to_exp_map", [a, ax], 3))
    exp_map_to_angle_axis = staticmethod(lambda e: oracle.arrays("exp_map_to_angle_axis", [e], (1, 3)))
    calc_heading = staticmethod(lambda q: OracleTJ._one("calc_heading", [q], 1))
    calc_heading_quat = staticmethod(lambda q: OracleTJ._one("calc_heading_quat", [q], 4))


@pytest.mark.parametrize("fixture", ["math2", "math2_nonfinite", "math2_extreme"])
@pytest.mark.parametrize("case", MULTI, ids=[c[0] for c in MULTI])
def test_remaining_functions(case, fixture):
    cid, fn, args, outk, classes, scale = case
    g = H.load_golden(fixture)
    got = getattr(OracleTJ, fn)(*[g[a] if isinstance(a, str) else a for a in args])
    check_outputs(H, got, g, outk, classes, scale, cid)


def test_torch_rand_float_from_recorded_draws():
    g = H.load_golden("math2")
    got = oracle.torch_rand_float(-0.2, 0.35, g["rand_u"])
    assert H.bit_equal_fraction(got, g["torch_rand_float"]) == 1.0


def test_extract_positions_on_the_reference_recording():
    """utils/torch_jit_utils.py:670-681, recorded from the reference's function (rows of 286 and 300 floats, a NaN and
    an Inf among the inputs): pure column selection, bit for bit."""
    g = H.load_golden("extract_positions")
    for name in "abc":
        got = oracle.extract_positions(g["in_" + name])
        assert got.shape == g["out_" + name].shape and H.bit_equal_fraction(got, g["out_" + name]) == 1.0, name
    x = g["in_c"]
    assert np.array_equal(oracle.extract_positions(x, 15).view(np.uint32),
                          x.reshape(33, 300)[:, :195].reshape(33, 15, 13)[:, :, :3].reshape(33, 45).view(np.uint32))
