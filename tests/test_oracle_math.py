"""CPU checker's torch_jit_utils functions against golden vectors produced by
the reference's own functions (tests/golden/math.npz)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H
from tests.math_cases import AFFINE, ROWWISE


@pytest.mark.parametrize("name,ins,outk,exact,scale", ROWWISE)
def test_rowwise(name, ins, outk, exact, scale):
    g = H.load_golden("math")
    got = oracle.rowwise(name, *[g[k] for k in ins])
    ref = g[outk]
    assert got.shape == ref.shape
    if exact:
        assert H.bit_equal_fraction(got, ref) == 1.0
    elif name == "normalize_angle":
        H.assert_angle_close(got, ref, what=name)
    else:
        H.assert_close(got, ref, scale=scale, what=name)


@pytest.mark.parametrize("name", AFFINE)
def test_affine(name):
    g = H.load_golden("math")
    got = oracle.affine(name, g["aff_x"], g["aff_lo"], g["aff_hi"])
    assert H.bit_equal_fraction(got, g[name]) == 1.0
