"""VecTask buffer contract and step tail: numpy restatement against the recording of the
reference's own base class (tests/golden/vectask.npz)."""
import numpy as np

from oracle import oracle
from tests import helpers as H


def test_step_tail_matches_reference():
    g = H.load_golden("vectask")
    for name, v in g["meta"]["variants"].items():
        clip = np.inf if v["clip"] is None else v["clip"]
        for t in range(g["meta"]["T"]):
            obs, tout = oracle.vec_task_tail(g[name + "_in_obs"][t], clip, g[name + "_in_progress"][t],
                                             g[name + "_in_reset"][t], v["max_len"])
            assert np.array_equal(obs.view(np.uint32), g[name + "_out_obs"][t].view(np.uint32))
            assert np.array_equal(tout, g[name + "_out_time_outs"][t])
            if v["num_states"]:
                st, _ = oracle.vec_task_tail(g[name + "_in_states"][t], clip, g[name + "_in_progress"][t],
                                             g[name + "_in_reset"][t], v["max_len"])
                assert np.array_equal(st.view(np.uint32), g[name + "_out_states"][t].view(np.uint32))
            acts, _ = oracle.vec_task_tail(g[name + "_in_actions"][t], 1.0, g[name + "_in_progress"][t],
                                           g[name + "_in_reset"][t], v["max_len"])
            assert np.array_equal(acts.view(np.uint32), g[name + "_clipped_actions"][t].view(np.uint32))


def test_buffer_contract_recorded():
    g = H.load_golden("vectask")
    a = lambda k: g["clip5_alloc_" + k]  # noqa: E731
    assert a("reset_buf").dtype == np.int64 and a("reset_buf").all()          # starts at ones
    for k in ("timeout_buf", "progress_buf", "randomize_buf"):
        assert a(k).dtype == np.int64 and not a(k).any()
    for k in ("obs_buf", "states_buf", "rew_buf"):
        assert a(k).dtype == np.float32 and not a(k).any()
