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
