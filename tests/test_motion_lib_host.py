"""Host side of the product MotionLib (no GPU): load-time tables and numpy-RNG
sampling against what the reference MotionLib produced (tests/golden/motion.npz)."""
import numpy as np
import torch
import pytest

from isaacgym_amd.tasks.amp.utils_amp.motion_lib import MotionLib, load_motion_file
from tests import helpers as H


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    g = H.load_golden("motion")
    path = H.write_golden_motions(g, str(tmp_path_factory.mktemp("motions")))
    return g, MotionLib(path, 28, g["meta"]["key_body_ids"], "cpu")


def test_load_time_tables_are_bit_identical(lib):
    g, ml = lib
    want = H.motion_tabs_from_golden(g)
    assert ml.num_motions() == 2
    for k, v in want.items():
        got = ml._tabs[k].numpy().reshape(v.shape)
        assert got.dtype == v.dtype, k
        assert np.array_equal(got.view(np.uint8), v.view(np.uint8)), k
    assert np.array_equal(ml._motion_weights, g["motion_weights"])
    assert ml.get_total_length() == float(np.sum(g["motion_lengths"]))


def test_sampling_draws_the_reference_sequence(lib):
    g, ml = lib
    np.random.seed(g["meta"]["np_seed"])
    ids = ml.sample_motions(600)
    times = ml.sample_time(ids)
    assert np.array_equal(ids, g["sampled_ids"])
    assert np.array_equal(times, g["sampled_times"])
    assert np.array_equal(ml.get_motion_length(ids), g["motion_lengths"][ids])


def test_frame_blend_host_copy(lib):
    g, ml = lib
    ids, times = g["ids"], g["times"]
    f0, f1, blend = ml._calc_frame_blend(times, g["motion_lengths"][ids], g["motion_num_frames"][ids],
                                         g["motion_dt"][ids])
    assert np.array_equal(f0, g["frame_idx0"]) and np.array_equal(f1, g["frame_idx1"])
    assert np.array_equal(blend, g["blend"])


def test_no_silent_cpu_path(lib):
    g, ml = lib
    with pytest.raises((ValueError, ImportError)):
        ml.get_motion_state(g["ids"][:4], g["times"][:4])


def _poselib_dict(F=5, J=15):
    rng = np.random.default_rng(0)
    from collections import OrderedDict
    arr = lambda a: OrderedDict(arr=a, context=dict(dtype=str(a.dtype)))
    q = rng.standard_normal((F, J, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    tree = OrderedDict(node_names=["b%d" % i for i in range(J)],
                       parent_indices=arr(np.array([-1, 0, 1, 1, 3, 4, 1, 6, 7, 0, 9, 10, 0, 12, 13], np.int64)),
                       local_translation=arr(rng.standard_normal((J, 3)).astype(np.float32)))
    return OrderedDict(rotation=arr(q), root_translation=arr(rng.standard_normal((F, 3)).astype(np.float32)),
                       global_velocity=arr(rng.standard_normal((F, J, 3)).astype(np.float32)),
                       global_angular_velocity=arr(rng.standard_normal((F, J, 3)).astype(np.float32)),
                       skeleton_tree=tree, is_local=True, fps=np.float64(30.0), __name__="SkeletonMotion")


def test_poselib_npy_is_read_without_numpy_pickle_loading(tmp_path):
    """A .npy written the way poselib writes motions (np.save of a dict) loads through the restricted
    unpickler and equals its .npz conversion."""
    from isaacgym_amd.tasks.amp.utils_amp.motion_lib import convert_npy, read_poselib_npy
    p = tmp_path / "m.npy"
    d = _poselib_dict()
    np.save(p, d)
    got = read_poselib_npy(str(p))
    assert np.array_equal(got["rotation"]["arr"], d["rotation"]["arr"]) and got["fps"] == 30.0
    m = load_motion_file(str(p))
    convert_npy(str(p), str(tmp_path / "m.npz"))
    m2 = load_motion_file(str(tmp_path / "m.npz"))
    assert torch.equal(m.rotation, m2.rotation) and m.parent_indices == m2.parent_indices and m.fps == m2.fps


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo should-never-run > /dev/null",))


def test_pickle_naming_any_other_global_is_rejected(tmp_path):
    import pickle
    from isaacgym_amd.tasks.amp.utils_amp.motion_lib import read_poselib_npy
    for payload in ({"rotation": _Evil()}, {"x": np.float32}, {"f": print}):
        p = tmp_path / "evil.npy"
        np.save(p, payload, allow_pickle=True)
        with pytest.raises(pickle.UnpicklingError, match="refusing"):
            read_poselib_npy(str(p))
        with pytest.raises(pickle.UnpicklingError):
            load_motion_file(str(p))
    p = tmp_path / "plain.npy"
    np.save(p, np.zeros(3))
    with pytest.raises(ValueError, match="not a poselib motion"):
        load_motion_file(str(p))
    with pytest.raises(ValueError):
        MotionLib(str(p), 27, [5, 8, 11, 14], "cpu")
