"""Table of the batched torch_jit_utils functions shared by the CPU and GPU
math tests: name, golden input keys, golden output key, exactness class.

"exact": only IEEE add/mul/div (and fused ops where the reference fuses): bit
equality is required.  Otherwise the value passes through sin/cos/acos/atan2/
sqrt of a math library (SLEEF in the reference's torch build, glibc in the CPU
checker, OCML on the GPU) and is held to the 1e-5 tolerance with the given
natural scale.
"""
import numpy as np

ROWWISE = [
    # name, input keys, output key, exact?, scale
    ("quat_mul", ("qa", "qb"), "quat_mul", True, 1.0),
    ("quat_conjugate", ("qa",), "quat_conjugate", True, 1.0),
    ("quat_rotate", ("qa", "v"), "quat_rotate", True, 1.0),
    ("quat_rotate_inverse", ("qa", "v"), "quat_rotate_inverse", True, 1.0),
    ("slerp", ("slerp_q0", "slerp_q1", "slerp_t"), "slerp", False, 1.0),
    ("quat_from_angle_axis", ("angle", "axis"), "quat_from_angle_axis", False, 1.0),
    ("normalize_angle", ("na_x",), "normalize_angle", False, np.pi),
    ("quat_to_exp_map", ("qe",), "quat_to_exp_map", False, np.pi),
    ("exp_map_to_quat", ("e",), "exp_map_to_quat", False, 1.0),
    ("quat_to_tan_norm", ("qa",), "quat_to_tan_norm", True, 1.0),
    ("calc_heading_quat_inv", ("qa",), "calc_heading_quat_inv", False, 1.0),
]
AFFINE = ["scale", "unscale", "tensor_clamp"]

# The remaining functions of the module (tests/golden/math2.npz).  Each case:
#   id, function name, arguments (golden keys; non-strings are passed as they are),
#   output keys, class per output: "exact" (bit equality), "tol" (1e-5 of the given
#   scale) or "angle" (1e-5 at scale pi, strict: a value 2 pi away from the reference is an error).
PI = float(np.pi)
MULTI = [
    ("normalize3", "normalize", ("v",), ("normalize3",), ("exact",), 1.0),
    ("normalize4", "normalize", ("raw4",), ("normalize4",), ("tol",), 1.0),
    ("quat_unit", "quat_unit", ("raw4",), ("quat_unit",), ("tol",), 1.0),
    ("quat_apply", "quat_apply", ("qa", "v"), ("quat_apply",), ("exact",), 1.0),
    ("tf_inverse", "tf_inverse", ("qa", "v"), ("tf_inverse_q", "tf_inverse_t"), ("exact", "exact"), 1.0),
    ("tf_apply", "tf_apply", ("qa", "v", "v2"), ("tf_apply",), ("exact",), 1.0),
    ("tf_vector", "tf_vector", ("qa", "v"), ("tf_vector",), ("exact",), 1.0),
    ("tf_combine", "tf_combine", ("qa", "v", "qb", "v2"), ("tf_combine_q", "tf_combine_t"), ("exact", "exact"), 1.0),
    ("get_basis_vector", "get_basis_vector", ("qa", "v"), ("get_basis_vector",), ("exact",), 1.0),
    ("copysign_pos", "copysign", (PI / 2.0, "cs_b"), ("copysign_pos",), ("exact",), 1.0),
    ("copysign_neg", "copysign", (-2.5, "cs_b"), ("copysign_neg",), ("exact",), 1.0),
    ("get_euler_xyz", "get_euler_xyz", ("qa",), ("euler_roll", "euler_pitch", "euler_yaw"), ("angle", "angle", "angle"), PI),
    ("quat_from_euler_xyz", "quat_from_euler_xyz", ("er", "ep", "ey"), ("quat_from_euler_xyz",), ("tol",), 1.0),
    ("euler_xyz_to_exp_map", "euler_xyz_to_exp_map", ("er", "ep", "ey"), ("euler_xyz_to_exp_map",), ("tol",), PI),
    ("heading_and_up_z", "compute_heading_and_up", ("qa", "isr", "to_target", "vec0", "vec1", 2),
     ("hu2_torso_quat", "hu2_up_proj", "hu2_heading_proj", "hu2_up_vec", "hu2_heading_vec"), ("exact",) * 5, 1.0),
    ("heading_and_up_y", "compute_heading_and_up", ("qa", "isr", "to_target", "vec0", "vec1", 1),
     ("hu1_torso_quat", "hu1_up_proj", "hu1_heading_proj", "hu1_up_vec", "hu1_heading_vec"), ("exact",) * 5, 1.0),
    ("compute_rot", "compute_rot", ("qb", "v", "v2", "targets", "to_target"),
     ("rot_vel_loc", "rot_angvel_loc", "rot_roll", "rot_pitch", "rot_yaw", "rot_angle_to_target"),
     ("exact", "exact", "angle", "angle", "angle", "angle"), PI),
    ("quat_axis0", "quat_axis", ("qa", 0), ("quat_axis0",), ("exact",), 1.0),
    ("quat_axis1", "quat_axis", ("qa", 1), ("quat_axis1",), ("exact",), 1.0),
    ("quat_axis2", "quat_axis", ("qa", 2), ("quat_axis2",), ("exact",), 1.0),
    ("scale_transform", "scale_transform", ("aff_x", "aff_lo", "aff_hi"), ("scale_transform",), ("exact",), 1.0),
    ("unscale_transform", "unscale_transform", ("aff_x", "aff_lo", "aff_hi"), ("unscale_transform",), ("exact",), 1.0),
    ("saturate", "saturate", ("aff_x", "aff_lo", "aff_hi"), ("saturate",), ("exact",), 1.0),
    ("quat_diff_rad", "quat_diff_rad", ("diff_a", "diff_b"), ("quat_diff_rad",), ("tol",), PI),
    ("local_to_world_space", "local_to_world_space", ("v", "pose"), ("local_to_world_space",), ("exact",), 1.0),
    ("quat_to_angle_axis", "quat_to_angle_axis", ("qe",), ("q2aa_angle", "q2aa_axis"), ("tol", "tol"), PI),
    ("angle_axis_to_exp_map", "angle_axis_to_exp_map", ("q2aa_angle", "q2aa_axis"), ("angle_axis_to_exp_map",), ("exact",), 1.0),
    ("exp_map_to_angle_axis", "exp_map_to_angle_axis", ("e",), ("e2aa_angle", "e2aa_axis"), ("tol", "tol"), PI),
    ("calc_heading", "calc_heading", ("qa",), ("calc_heading",), ("angle",), PI),
    ("calc_heading_quat", "calc_heading_quat", ("qa",), ("calc_heading_quat",), ("tol",), 1.0),
]


def well_conditioned_rows(name, g):
    """Rows of a ROWWISE case whose result is a function of the inputs rather than of the last bit of a library
    function.  slerp evaluates sin(t * acos(c)): with |t| beyond a few units an ulp of acos (on which the math
    libraries legitimately differ) is a large change of the argument - at t = 1e30 the sine is arbitrary, in the
    reference as much as anywhere.  Rows with a finite |t| > 16 are left out (only the extreme-value recording has
    any); non-finite t stays in (sin(Inf) is NaN everywhere)."""
    if name != "slerp":
        return slice(None)
    t = np.abs(g["slerp_t"].reshape(-1))
    return ~(np.isfinite(t) & (t > 16.0))


def check_outputs(H, got, g, outk, classes, scale, what):
    """Compare the tuple `got` with the golden arrays named by outk."""
    if not isinstance(got, tuple):
        got = (got,)
    assert len(got) == len(outk), what
    for x, k, cls in zip(got, outk, classes):
        ref = g[k]
        x = np.asarray(x)
        assert x.shape == ref.shape, "%s %s: shape %s, want %s" % (what, k, x.shape, ref.shape)
        if cls == "exact":
            assert H.bit_equal_fraction(x, ref) == 1.0, "%s %s: bit-equal fraction %f" % (what, k, H.bit_equal_fraction(x, ref))
        elif cls == "angle":
            H.assert_angle_close(x, ref, what="%s %s" % (what, k))
        else:
            H.assert_close(x, ref, scale=scale, what="%s %s" % (what, k))
