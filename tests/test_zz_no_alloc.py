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
        # torch's counter only sees torch's own allocator; a hipMalloc inside the library would show up in the
        # device's free memory instead (hipMemGetInfo), so that is compared around a second, warm call
        warm = {k: (v.clone() if v is not None else None) for k, v in got.items()}
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        assert _call("igx_allegro_post_physics_static", p, warm, n) == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the device's free memory changed across the call"
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


def test_static_entry_point_captured_and_replayed():
    """After igx_allegro_static_prepare() the entry point only enqueues kernels: it is recorded into a HIP graph
    and replayed on new inputs, and every replay leaves what the workspace entry point leaves."""
    from isaacgym_amd import ops
    n = 30011
    p = synthetic.allegro_params("full_state", seed=5, rng_offset=2)
    ops.allegro_static_prepare(DEV)
    ref, got = _pair(n, 21)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up outside the capture, as torch's graph recipe asks
        assert _call("igx_allegro_post_physics_static", p, {k: (v.clone() if v is not None else None) for k, v in got.items()}, n) == 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert _call("igx_allegro_post_physics_static", p, got, n) == 0
    for replay in range(3):
        fresh = synthetic.allegro_state(n, 30 + replay, DEV, "full_state", False)
        fresh["consecutive_successes"].fill_(0.25 * replay)
        for k, v in fresh.items():
            if v is not None and k != "workspace":
                got[k].copy_(v)
                ref[k].copy_(v)
        graph.replay()
        assert _call("igx_allegro_post_physics", p, ref, n) == 0
        torch.cuda.synchronize()
        for k in OUT:
            assert torch.equal(ref[k].view(torch.uint8), got[k].view(torch.uint8)), "%s (replay %d)" % (k, replay)


def test_first_call_under_capture_is_refused_in_a_fresh_process():
    """Without the set-up call, a first call per device that arrives under capture must not look symbols up
    inside the capture: it fails with the remedy named.  (Own process: the address is resolved once per process.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from isaacgym_amd import _abi, _lib, synthetic
p = synthetic.allegro_params("full_state")
t = synthetic.allegro_state(4096, 1, "cuda:0", "full_state", False)
t["workspace"] = None
b, dev = _lib.make_buffers(_abi.AllegroBuffers, t)
torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    rc = _lib.lib().igx_allegro_post_physics_static(C.byref(p), C.byref(b), C.c_int64(4096), _lib.current_stream(dev))
msg = _lib.lib().igx_last_error_string().decode()
assert rc == _abi.IGX_E_PARAM, (rc, msg)
assert "igx_allegro_static_prepare" in msg, msg
assert _lib.lib().igx_allegro_static_prepare() == 0
assert _lib.lib().igx_allegro_post_physics_static(C.byref(p), C.byref(b), C.c_int64(4096), _lib.current_stream(dev)) == 0
torch.cuda.synchronize()
print("refused under capture, fine after prepare")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "refused under capture" in r.stdout, (r.stdout, r.stderr[-3000:])
