"""Episode-statistics bookkeeping: CPU checker against the torch evaluation of the reference's
call site (learning/common_agent.py:284-296; tests/golden/episode_stats.npz)."""
import numpy as np

from oracle import oracle
from tests import helpers as H


def test_episode_stats_oracle_matches_call_site():
    g = H.load_golden("episode_stats")
    N, T = g["meta"]["N"], g["meta"]["T"]
    cr, cl = np.zeros(N, np.float32), np.zeros(N, np.float32)
    for t in range(T):
        n, sr, sl = oracle.episode_stats_step(dict(rew=g["rew"][t].copy(), done=g["dones"][t].copy(),
                                                   current_rewards=cr, current_lengths=cl,
                                                   workspace=np.zeros(8, np.int64)), N)
        assert n == int(g["n_done"][t])
        assert np.array_equal(cr.view(np.uint32), g["current_rewards"][t].view(np.uint32))  # incl. -0.0
        assert np.array_equal(cl.view(np.uint32), g["current_lengths"][t].view(np.uint32))
        assert sr == float(g["sum_reward"][t]) and sl == float(g["sum_length"][t])
        if n:
            np.testing.assert_allclose(sr / n, g["mean_reward"][t], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(sl / n, g["mean_length"][t], rtol=1e-6)
