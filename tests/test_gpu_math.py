"""GPU parity of the batched torch_jit_utils kernels: against the reference's
golden vectors and against the CPU checker on larger seeded inputs."""
import numpy as np
import pytest
import torch

from isaacgym_amd.utils import torch_jit_utils as tj
from oracle import oracle
from tests import helpers as H
from tests.math_cases import AFFINE, ROWWISE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _call(name, args):
    fn = getattr(tj, name)
    if name in ("quat_from_angle_axis",):
        return fn(args[0], args[1])
    if name == "normalize_angle":
        return fn(args[0])
    return fn(*args)


@pytest.mark.parametrize("name,ins,outk,exact,scale", ROWWISE)
def test_rowwise_golden(name, ins, outk, exact, scale):
    g = H.load_golden("math")
    got = _call(name, [_dev(g[k]) for k in ins]).cpu().numpy()
    ref = g[outk]
    assert got.shape == ref.shape
    if exact:
        assert H.bit_equal_fraction(got, ref) == 1.0
    elif name == "normalize_angle":
        H.assert_angle_close(got, ref, what=name)
    else:
        H.assert_close(got, ref, scale=scale, what=name)


@pytest.mark.parametrize("name", AFFINE)
def test_affine_golden(name):
    g = H.load_golden("math")
    got = getattr(tj, name)(_dev(g["aff_x"]), _dev(g["aff_lo"]), _dev(g["aff_hi"])).cpu().numpy()
    assert H.bit_equal_fraction(got, g[name]) == 1.0


@pytest.mark.parametrize("n", [0, 1, 255, 257, 300001])
def test_rowwise_vs_oracle_sizes(n):
    gen = torch.Generator(device=DEV).manual_seed(n + 5)
    qa = torch.randn(n, 4, generator=gen, device=DEV)
    qb = torch.randn(n, 4, generator=gen, device=DEV)
    qa = qa / qa.norm(dim=-1, keepdim=True).clamp(min=1e-9)
    qb = qb / qb.norm(dim=-1, keepdim=True).clamp(min=1e-9)
    v = torch.randn(n, 3, generator=gen, device=DEV)
    t = torch.rand(n, 1, generator=gen, device=DEV) * 2 - 0.5
    cases = [("quat_mul", (qa, qb), True), ("quat_rotate", (qa, v), True), ("quat_rotate_inverse", (qa, v), True),
             ("quat_conjugate", (qa,), True), ("slerp", (qa, qb, t), False), ("quat_to_tan_norm", (qa,), True),
             ("quat_to_exp_map", (qa,), False), ("exp_map_to_quat", (v,), False),
             ("calc_heading_quat_inv", (qa,), False)]
    for name, args, exact in cases:
        got = getattr(tj, name)(*args).cpu().numpy()
        if n == 0:
            assert got.shape[0] == 0
            continue
        ref = oracle.rowwise(name, *[a.cpu().numpy() for a in args])
        if exact:
            assert H.bit_equal_fraction(got, ref) == 1.0, name
        else:
            H.assert_close(got, ref, scale=np.pi if name == "quat_to_exp_map" else 1.0, what=name)


def test_shapes_follow_the_reference():
    q = torch.randn(5, 15, 4, device=DEV)
    q = q / q.norm(dim=-1, keepdim=True)
    t = torch.rand(5, 1, 1, device=DEV)
    assert tj.slerp(q, q.flip(0), t).shape == (5, 15, 4)
    assert tj.quat_mul(q, q).shape == (5, 15, 4)
    assert tj.quat_to_exp_map(q).shape == (5, 15, 3)
    assert tj.quat_to_tan_norm(q[:, 0]).shape == (5, 6)
