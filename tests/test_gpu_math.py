"""GPU parity of the batched torch_jit_utils kernels: against the reference's
golden vectors and against the CPU checker on larger seeded inputs."""
import numpy as np
import pytest
import torch

from isaacgym_amd.utils import torch_jit_utils as tj
from oracle import oracle
from tests import helpers as H
from tests.math_cases import AFFINE, MULTI, ROWWISE, check_outputs, well_conditioned_rows

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


@pytest.mark.parametrize("fixture", ["math", "math_nonfinite", "math_extreme"])
@pytest.mark.parametrize("name,ins,outk,exact,scale", ROWWISE)
def test_rowwise_golden(name, ins, outk, exact, scale, fixture):
    g = H.load_golden(fixture)
    got = _call(name, [_dev(g[k]) for k in ins]).cpu().numpy()
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
def test_affine_golden(name, fixture):
    g = H.load_golden(fixture)
    got = getattr(tj, name)(_dev(g["aff_x"]), _dev(g["aff_lo"]), _dev(g["aff_hi"])).cpu().numpy()
    assert H.bit_equal_fraction(got, g[name]) == 1.0


@pytest.mark.parametrize("n,d,offset", [(1, 3, 0), (257, 7, 0), (4099, 8, 0), (4099, 8, 1), (100003, 21, 0), (65536, 20, 0)])
def test_affine_vs_oracle_shapes(n, d, offset):
    """Both launch shapes of the affine kernels (four elements per lane when the element count is a multiple of
    four and the bases are 16-byte aligned, one per lane otherwise), bit for bit against the CPU checker."""
    gen = torch.Generator(device=DEV).manual_seed(n + d)
    flat = torch.randn(n * d + offset, generator=gen, device=DEV)
    x = flat[offset:].view(n, d)  # offset 1: a base that is only 4-byte aligned
    lo = -torch.rand(d, generator=gen, device=DEV) - 0.1
    hi = torch.rand(d, generator=gen, device=DEV) + 0.1
    for name in AFFINE:
        got = getattr(tj, name)(x, lo, hi).cpu().numpy()
        ref = oracle.affine(name, x.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy())
        assert H.bit_equal_fraction(got, ref) == 1.0, name


def test_affine_bounds_per_element_and_strided_inputs():
    """Bounds of the array's full shape (or any shape that broadcasts to it) and inputs that are views with
    strides: the reference's expressions accept both (utils/torch_jit_utils.py:228-241)."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(301, 7, generator=gen, device=DEV)
    lo = -torch.rand(301, 7, generator=gen, device=DEV) - 0.1
    hi = torch.rand(301, 1, generator=gen, device=DEV) + 0.1
    for name in AFFINE:
        got = getattr(tj, name)(x, lo, hi).cpu().numpy()
        ref = oracle.affine(name, x.cpu().numpy().reshape(1, -1), lo.cpu().numpy().reshape(-1),
                            hi.expand(301, 7).cpu().numpy().reshape(-1)).reshape(301, 7)
        assert H.bit_equal_fraction(got, ref) == 1.0, name
    wide = torch.randn(301, 14, generator=gen, device=DEV)
    view, bounds = wide[:, ::2], torch.randn(2, 14, generator=gen, device=DEV).sort(dim=0).values[:, ::2]
    for name in AFFINE:
        got = getattr(tj, name)(view, bounds[0], bounds[1]).cpu().numpy()
        ref = oracle.affine(name, view.contiguous().cpu().numpy(), bounds[0].contiguous().cpu().numpy(),
                            bounds[1].contiguous().cpu().numpy())
        assert H.bit_equal_fraction(got, ref) == 1.0, name
    q = torch.randn(100, 8, generator=gen, device=DEV)
    a, b = q[:, :4], q[:, 4:]  # rows 32 bytes apart
    got = tj.quat_mul(a, b).cpu().numpy()
    assert H.bit_equal_fraction(got, oracle.rowwise("quat_mul", a.contiguous().cpu().numpy(), b.contiguous().cpu().numpy())) == 1.0


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


def test_float64_inputs_are_refused_not_narrowed():
    q = torch.zeros(3, 4, dtype=torch.float64, device=DEV)
    with pytest.raises(TypeError):
        tj.quat_conjugate(q)
    with pytest.raises(TypeError):
        tj.scale(q, q[0], q[0])
    assert tj.quat_conjugate(q.half()).dtype == torch.float32  # narrower types widen, as torch's promotion would


def test_shapes_follow_the_reference():
    q = torch.randn(5, 15, 4, device=DEV)
    q = q / q.norm(dim=-1, keepdim=True)
    t = torch.rand(5, 1, 1, device=DEV)
    assert tj.slerp(q, q.flip(0), t).shape == (5, 15, 4)
    assert tj.quat_mul(q, q).shape == (5, 15, 4)
    assert tj.quat_to_exp_map(q).shape == (5, 15, 3)
    assert tj.quat_to_tan_norm(q[:, 0]).shape == (5, 6)


@pytest.mark.parametrize("fixture", ["math2", "math2_nonfinite", "math2_extreme"])
@pytest.mark.parametrize("case", MULTI, ids=[c[0] for c in MULTI])
def test_remaining_functions_golden(case, fixture):
    cid, fn, args, outk, classes, scale = case
    g = H.load_golden(fixture)
    got = getattr(tj, fn)(*[_dev(g[a]) if isinstance(a, str) else a for a in args])
    got = tuple(t.cpu().numpy() for t in (got if isinstance(got, tuple) else (got,)))
    check_outputs(H, got, g, outk, classes, scale, cid)


@pytest.mark.parametrize("n", [0, 1, 255, 257, 100003])
def test_remaining_functions_vs_oracle_sizes(n):
    """Sizes around the 256-row tile, an empty batch and a large odd one, against the CPU checker."""
    from tests.test_oracle_math import OracleTJ
    gen = torch.Generator(device=DEV).manual_seed(n + 11)

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True).clamp(min=1e-9)

    qa, qb = unit(torch.randn(n, 4, generator=gen, device=DEV)), unit(torch.randn(n, 4, generator=gen, device=DEV))
    v, w, x = (torch.randn(n, 3, generator=gen, device=DEV) for _ in range(3))
    pose = torch.cat([x, qb], dim=-1)
    cases = [("quat_apply", (qa, v), ("exact",)), ("tf_inverse", (qa, v), ("exact", "exact")),
             ("tf_combine", (qa, v, qb, w), ("exact", "exact")), ("tf_apply", (qa, v, w), ("exact",)),
             ("compute_heading_and_up", (qa, qb, v, w, x, 2), ("exact",) * 5),
             ("compute_rot", (qa, v, w, x, v), ("exact", "exact", "angle", "angle", "angle", "angle")),
             ("local_to_world_space", (v, pose), ("exact",)), ("quat_axis", (qa, 1), ("exact",)),
             ("normalize", (v,), ("exact",)), ("get_euler_xyz", (qa,), ("angle",) * 3),
             ("quat_diff_rad", (qa, qb), ("tol",)), ("calc_heading_quat", (qa,), ("tol",))]
    for name, args, classes in cases:
        got = getattr(tj, name)(*args)
        got = tuple(t.cpu().numpy() for t in (got if isinstance(got, tuple) else (got,)))
        if n == 0:
            assert all(t.shape[0] == 0 for t in got), name
            continue
        ref = getattr(OracleTJ, name)(*[a.cpu().numpy() if torch.is_tensor(a) else a for a in args])
        ref = ref if isinstance(ref, tuple) else (ref,)
        for k, (a, b, cls) in enumerate(zip(got, ref, classes)):
            b = b.reshape(a.shape)
            if cls == "exact":
                assert H.bit_equal_fraction(a, b) == 1.0, (name, k)
            elif cls == "angle":
                H.assert_angle_close(a, b, what=name)
            else:
                H.assert_close(a, b, scale=np.pi, what=name)


def test_torch_rand_float():
    g = H.load_golden("math2")
    u = _dev(g["rand_u"])
    got = tj.torch_rand_float(-0.2, 0.35, tuple(u.shape), DEV, uniforms=u)
    assert H.bit_equal_fraction(got.cpu().numpy(), g["torch_rand_float"]) == 1.0
    # without recorded draws: the library's own counter-based stream, the same one the CPU checker restates
    got = tj.torch_rand_float(-1.0, 3.0, (1001, 7), DEV, seed=77, rng_offset=5)
    ref = oracle.torch_rand_float(-1.0, 3.0, None, seed=77, rng_offset=5, count=1001 * 7)
    assert np.array_equal(got.cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32))
    assert float(got.min()) >= -1.0 and float(got.max()) < 3.0 and abs(float(got.mean()) - 1.0) < 0.1
    other = tj.torch_rand_float(-1.0, 3.0, (1001, 7), DEV, seed=77, rng_offset=6)
    assert not torch.equal(got, other)


def test_extract_positions_golden_and_sizes():
    """extract_positions_from_rigid_body_states (utils/torch_jit_utils.py:670-681): the reference's recording (rows of
    286 and of 300 floats), then batch sizes around the 16-row tile, other body counts, unaligned bases and wide rows
    against the CPU checker; pure column selection, bit for bit."""
    g = H.load_golden("extract_positions")
    for name in "abc":
        got = tj.extract_positions_from_rigid_body_states(_dev(g["in_" + name])).cpu().numpy()
        assert got.shape == g["out_" + name].shape and H.bit_equal_fraction(got, g["out_" + name]) == 1.0, name
    rng = np.random.default_rng(5)
    for n, width, bodies, offset in ((0, 286, 22, 0), (1, 286, 22, 0), (15, 286, 22, 0), (16, 286, 22, 0), (17, 286, 22, 0),
                                     (1000, 286, 22, 0), (100003, 286, 22, 0), (333, 195, 15, 0), (333, 300, 22, 0),
                                     (50, 13, 1, 0), (257, 416, 32, 0), (100, 286, 22, 1)):
        flat = rng.standard_normal(n * width + offset).astype(np.float32)
        x = _dev(flat)[offset:].view(n, width)  # offset 1: a base that is not 16-byte aligned
        got = tj.extract_positions_from_rigid_body_states(x, bodies).cpu().numpy()
        want = oracle.extract_positions(flat[offset:].reshape(n, width), bodies)
        assert got.shape == (n, 3 * bodies) and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, width, bodies)
    with pytest.raises(IndexError):
        tj.extract_positions_from_rigid_body_states(torch.zeros(4, 285, device=DEV))
