"""EpisodeStats on the GPU: golden replay and large-n comparison with the CPU checker."""
import numpy as np
import pytest
import torch

from isaacgym_amd.episode_stats import EpisodeStats
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_episode_stats_golden():
    g = H.load_golden("episode_stats")
    N, T = g["meta"]["N"], g["meta"]["T"]
    es = EpisodeStats(N, DEV)
    tot_n, tot_r, tot_l = 0, 0.0, 0.0
    for t in range(T):
        es.step(torch.from_numpy(g["rew"][t]).to(DEV), torch.from_numpy(g["dones"][t]).to(DEV))
        n, sr, sl = es.last_step()
        assert n == int(g["n_done"][t])
        np.testing.assert_allclose(sr, float(g["sum_reward"][t]), rtol=1e-6, atol=1e-5)
        assert sl == float(g["sum_length"][t])  # sums of small integers are exact
        assert np.array_equal(es.current_rewards.cpu().numpy().view(np.uint32), g["current_rewards"][t].view(np.uint32))
        assert np.array_equal(es.current_lengths.cpu().numpy().view(np.uint32), g["current_lengths"][t].view(np.uint32))
        tot_n += n; tot_r += sr; tot_l += sl
    n, sr, sl, steps = es.totals()
    assert (n, steps) == (tot_n, T) and sl == tot_l
    np.testing.assert_allclose(sr, tot_r, rtol=1e-9)
    assert es.totals() == (0, 0.0, 0.0, 0)  # reset by the read
    assert es.means() is None


@pytest.mark.parametrize("n", [1, 255, 256, 100003, 1 << 20])
def test_episode_stats_vs_oracle(n):
    es = EpisodeStats(n, DEV)
    gen = torch.Generator(device=DEV).manual_seed(n)
    cr, cl = np.zeros(n, np.float32), np.zeros(n, np.float32)
    tot = [0, 0.0, 0.0]
    for t in range(3):
        rew = torch.randn(n, generator=gen, device=DEV)
        done = (torch.rand(n, generator=gen, device=DEV) < 0.1).to(torch.int64)
        es.step(rew, done)
        k, sr, sl = oracle.episode_stats_step(dict(rew=rew.cpu().numpy(), done=done.cpu().numpy(), current_rewards=cr,
                                                   current_lengths=cl, workspace=np.zeros(8, np.int64)), n)
        got = es.last_step()
        assert got[0] == k and got[2] == sl
        np.testing.assert_allclose(got[1], sr, rtol=1e-6, atol=1e-4)
        assert np.array_equal(es.current_rewards.cpu().numpy().view(np.uint32), cr.view(np.uint32))
        assert np.array_equal(es.current_lengths.cpu().numpy().view(np.uint32), cl.view(np.uint32))
        tot[0] += k; tot[1] += sr; tot[2] += sl
    m = es.means()
    if tot[0]:
        np.testing.assert_allclose(m, (tot[1] / tot[0], tot[2] / tot[0]), rtol=1e-6, atol=1e-5)
