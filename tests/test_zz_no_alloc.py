"""igx_allegro_post_physics_static: the AllegroHand post-physics step for callers that cannot
allocate (its reduction records live in a file-scope device array inside libigx.so).  Called
through the C ABI and compared with igx_allegro_post_physics on the same inputs: flags and
observations bit for bit, reward / consecutive_successes bit for bit too (same kernels, same
reduction order) - twice in a row, the second call on fewer environments, so the records the
first call left behind in the shared array must not leak into the second."""
import ctypes as C

import pytest
import torch

from isaacgym_amd import _abi, _lib, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OUT = ("obs", "rew", "reset", "reset_goal", "progress", "randomize", "successes", "consecutive_successes")


def _call(symbol, p, t, n):
    b, dev = _lib.make_buffers(_abi.AllegroBuffers, t)
    with torch.cuda.device(dev):
        return getattr(_lib.lib(), symbol)(C.byref(p), C.byref(b), C.c_int64(n), _lib.current_stream(dev))


def _pair(n, seed):
    a = synthetic.allegro_state(n, seed, DEV, "full_state", False)
    a["consecutive_successes"].fill_(0.5)
    b = {k: (v.clone() if v is not None else None) for k, v in a.items()}
    b["workspace"] = None  # the static entry point must not need it
    return a, b


def test_static_entry_point_equals_the_workspace_one_twice_in_a_row():
    p = synthetic.allegro_params("full_state", seed=3, rng_offset=1)
    for n, seed in ((200003, 11), (4099, 12)):  # the second call is smaller: stale records beyond its tiles
        ref, got = _pair(n, seed)
        assert _call("igx_allegro_post_physics", p, ref, n) == 0
        before = torch.cuda.memory_allocated()
        assert _call("igx_allegro_post_physics_static", p, got, n) == 0
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        assert int(ref["reset"].sum()) > 0  # the reduction has something to add up
        for k in OUT:
            if ref[k] is None:
                continue
            assert torch.equal(ref[k].view(torch.uint8), got[k].view(torch.uint8)), "%s (n=%d)" % (k, n)


def test_static_entry_point_rejects_what_does_not_fit():
    p = synthetic.allegro_params("full_state")
    t = synthetic.allegro_state(64, 1, DEV, "full_state", False)
    rc = _call("igx_allegro_post_physics_static", p, t, _abi.IGX_ALLEGRO_STATIC_MAX_ENVS + 1)
    assert rc == _abi.IGX_E_SIZE
    assert b"IGX_ALLEGRO_STATIC_MAX_ENVS" in _lib.lib().igx_last_error_string()
    assert _call("igx_allegro_post_physics_static", p, t, 0) == 0
