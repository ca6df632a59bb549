"""Host side of the product MotionLib (no GPU): load-time tables and numpy-RNG
sampling against what the reference MotionLib produced (tests/golden/motion.npz)."""
import numpy as np
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


def test_pickled_motion_needs_opt_in(tmp_path):
    p = tmp_path / "m.npy"
    np.save(p, np.zeros(3))
    with pytest.raises(ValueError, match="allow_pickle"):
        load_motion_file(str(p))
    with pytest.raises(ValueError):
        MotionLib(str(p), 27, [5, 8, 11, 14], "cpu")
