"""Host threads calling the library at once, each on its own stream and buffers: the C ABI keeps no state between
calls apart from per-thread error text and constants read once, so the results must be those of the same calls made
one after the other (ctypes releases the interpreter lock for the duration of a foreign call, so the library's host
code really does run concurrently here)."""
import threading

import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _work(kind, seed, n, steps, stream):
    """`steps` launches of one task's step on its own buffers; returns the buffers that the steps wrote."""
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        if kind == "ant":
            p, t = synthetic.loco_params(synthetic.ANT, seed=seed, rng_offset=0), synthetic.loco_state(synthetic.ANT, n, seed, DEV)
            step, keys = ops.ant_step, ("obs", "rew", "reset", "progress", "dof_state", "potentials")
        elif kind == "humanoid":
            p, t = synthetic.loco_params(synthetic.HUMANOID, seed=seed, rng_offset=0), synthetic.loco_state(synthetic.HUMANOID, n, seed, DEV)
            step, keys = ops.humanoid_step, ("obs", "rew", "reset", "progress", "dof_state", "potentials")
        elif kind == "allegro":
            p, t = synthetic.allegro_params("full_state", seed, 0), synthetic.allegro_state(n, seed, DEV)
            step, keys = ops.allegro_post_physics, ("obs", "rew", "reset", "progress", "successes", "consecutive_successes")
        else:
            p, t = synthetic.amp_params(2), synthetic.amp_state(n, seed, DEV, 2)
            step, keys = ops.humanoid_amp_step, ("obs", "rew", "reset", "progress", "amp_obs")
        cache = {}
        for i in range(steps):
            if hasattr(p, "rng_offset"):
                p.rng_offset = i
            step(p, t, n, cache)
        (stream or torch.cuda.current_stream()).synchronize()
        return {k: t[k].cpu().numpy().copy() for k in keys if k in t}


def test_four_host_threads_on_four_streams_equal_the_serial_calls():
    jobs = [("ant", 11, 20001, 40), ("humanoid", 12, 9001, 40), ("allegro", 13, 7001, 40), ("amp", 14, 5001, 40)]
    serial = [_work(kind, seed, n, steps, None) for kind, seed, n, steps in jobs]
    results, errors = [None] * len(jobs), []

    def run(i):
        try:
            torch.cuda.set_device(0)
            results[i] = _work(*jobs[i], torch.cuda.Stream(device=DEV))
        except Exception as e:  # surfaced below: an exception in a thread must fail the test
            errors.append((jobs[i][0], repr(e)))

    for _ in range(3):  # three rounds of four concurrent threads
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=120)
        assert not errors, errors
        for (kind, *_), got, want in zip(jobs, results, serial):
            assert got is not None, kind
            for k in want:
                assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), (kind, k)
