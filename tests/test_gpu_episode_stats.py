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


def test_observer_matches_the_reference_run():
    """observer.npz: the reference's CommonAgent.play_steps + RLGPUAlgoObserver; here EpisodeStats and the
    product's observer on the device (igx_episode_stats_step, igx_episode_window_step)."""
    from isaacgym_amd.utils.rlgames_utils import RLGPUAlgoObserver
    g = H.load_golden("observer")
    es = EpisodeStats(g["meta"]["N"], DEV)

    def stats_step(rew, dones):
        es.step(rew, dones)
        n, sr, sl = es.last_step()
        return n, sr, sl, es.current_rewards, es.current_lengths

    H.observer_replay(g, RLGPUAlgoObserver(), stats_step, to_tensor=lambda a: torch.tensor(np.asarray(a)).to(DEV),
                      to_np=lambda t: t.cpu().numpy(), sum_rtol=1e-12)


@pytest.mark.parametrize("n,G,p", [(1, 4, 1.0), (1023, 100, 0.3), (1024, 7, 0.9), (70001, 128, 0.01), (1 << 20, 1000, 0.02)])
def test_episode_window_vs_oracle(n, G, p):
    """igx_episode_window_step against the CPU checker: accumulators, rings and counters bit for bit, with
    steps that finish fewer and more episodes than the window holds, and the index-based entry point."""
    import types
    from isaacgym_amd.utils.rlgames_utils import RLGPUAlgoObserver
    obs = RLGPUAlgoObserver()
    obs.after_init(types.SimpleNamespace(games_to_track=G, device=DEV, writer=H.RecordingWriter()))
    gen = torch.Generator().manual_seed(n + G)
    keys = ["a", "b", "c"]
    cum = [np.zeros(n, np.float32) for _ in keys]
    win, ws = np.zeros((len(keys), G), np.float32), np.zeros(8 + (n + 1023) // 1024, np.int64)
    for t in range(4):
        vals = [torch.randn(n, generator=gen) for _ in keys]
        done = (torch.rand(n, generator=gen) < (p if t != 2 else 0.0)).to(torch.int64)
        infos = {"episode_cumulative": {k: v.to(DEV) for k, v in zip(keys, vals)}}
        if t == 1:  # the reference's signature: indices from nonzero()
            obs.process_infos(infos, done.nonzero(as_tuple=False).to(DEV))
        else:
            obs.process_infos_flags(infos, done.to(DEV))
        oracle.episode_window_step(done.numpy(), [v.numpy() for v in vals], cum, win, ws, G)
        for j, k in enumerate(keys):
            assert np.array_equal(obs.episode_cumulative[k].cpu().numpy().view(np.uint32), cum[j].view(np.uint32)), (k, t)
        assert obs._ws[:3].cpu().tolist() == ws[:3].tolist()
        total = int(ws[0])
        valid = [(i % G) for i in range(max(total - G, 0), total)]  # ring slots that hold an episode
        assert np.array_equal(obs._window.cpu().numpy()[:, valid].view(np.uint32), win[:, valid].view(np.uint32)), t


def test_game_stats_windowed_meters():
    """GameStats: per-step records replayed into rl_games-style windowed means once per horizon; equal to
    updating the meters with current_rewards[done_indices] every step (learning/common_agent.py:284-296)."""
    from isaacgym_amd.utils.rlgames_utils import AverageMeter, GameStats
    n = 5000
    gs = GameStats(n, DEV, games_to_track=100, horizon=4)
    ref_r, ref_l = AverageMeter(100), AverageMeter(100)
    cr, cl = torch.zeros(n), torch.zeros(n)
    gen = torch.Generator().manual_seed(5)
    for t in range(10):
        rew = torch.randn(n, generator=gen)
        dones = (torch.rand(n, generator=gen) < 0.004).to(torch.int64)
        gs.step(rew.to(DEV), dones.to(DEV))
        cr += rew
        cl += 1
        idx = dones.nonzero(as_tuple=False).squeeze(-1)
        ref_r.update(cr[idx])
        ref_l.update(cl[idx])
        cr *= 1.0 - dones.float()
        cl *= 1.0 - dones.float()
    gs.flush()
    assert len(gs.game_rewards) == len(ref_r) and len(gs.game_lengths) == len(ref_l)
    np.testing.assert_allclose(gs.game_rewards.get_mean(), ref_r.get_mean(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.game_lengths.get_mean(), ref_l.get_mean(), rtol=1e-6)
