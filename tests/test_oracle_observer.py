"""Episode statistics against the reference running: tests/golden/observer.npz is a recording of the
reference's own CommonAgent.play_steps loop (learning/common_agent.py:250-310) with its
RLGPUAlgoObserver attached (utils/rlgames_utils.py:132-211).  Here the CPU checker takes the place of
the device kernels under the product's host class, so the host logic (window order, tags, means,
min / max, direct scalars, 'no new episodes: no rows') is covered without a GPU."""
import numpy as np
import torch

from isaacgym_amd.utils.rlgames_utils import AverageMeter, RLGPUAlgoObserver
from oracle import oracle
from tests import helpers as H


class CheckerObserver(RLGPUAlgoObserver):
    """the product's observer with the CPU restatement of igx_episode_window_step under it"""

    def _window_step(self, cumulative, dones):
        keys = list(cumulative.keys())
        n = dones.shape[0]
        G = int(self.algo.games_to_track)
        if not self._keys:
            self._keys = keys
            for k in keys:
                self.episode_cumulative[k] = torch.zeros(n, dtype=torch.float32)
            self._window = torch.zeros(len(keys), G, dtype=torch.float32)
            self._ws = torch.zeros(64, dtype=torch.int64)
        oracle.episode_window_step(dones.numpy(), [cumulative[k].numpy() for k in keys],
                                   [self.episode_cumulative[k].numpy() for k in keys], self._window.numpy(),
                                   self._ws.numpy(), G)


def test_observer_and_agent_bookkeeping_match_the_reference_run():
    g = H.load_golden("observer")
    N = g["meta"]["N"]
    cr, cl = np.zeros(N, np.float32), np.zeros(N, np.float32)

    def stats_step(rew, dones):
        n, sr, sl = oracle.episode_stats_step(dict(rew=rew.numpy().copy(), done=dones.numpy().copy(), current_rewards=cr,
                                                   current_lengths=cl, workspace=np.zeros(8, np.int64)), N)
        return n, sr, sl, torch.from_numpy(cr), torch.from_numpy(cl)

    H.observer_replay(g, CheckerObserver(), stats_step, to_tensor=lambda a: torch.tensor(np.asarray(a)),
                      to_np=lambda t: t.numpy())


def test_window_keeps_the_last_games_of_a_large_step():
    """More finished episodes in one step than the window holds: the deque keeps the last G, in index order."""
    n, G = 50, 8
    done = np.ones(n, np.int64)
    vals = [np.arange(n, dtype=np.float32)]
    cum = [np.zeros(n, np.float32)]
    win, ws = np.zeros((1, G), np.float32), np.zeros(16, np.int64)
    oracle.episode_window_step(done, vals, cum, win, ws, G)
    total = int(ws[0])
    assert total == n and not cum[0].any()
    order = [(i % G) for i in range(total - G, total)]
    assert win[0, order].tolist() == list(range(n - G, n))


def test_average_meter_update_rule():
    """rl_games' windowed mean (restated: rl_games is not in the reference tree - parity unpinned)."""
    m = AverageMeter(4)
    m.update(torch.tensor([1.0, 3.0]))
    assert len(m) == 2 and m.get_mean() == 2.0
    m.update(torch.tensor([]))
    assert len(m) == 2
    m.update(torch.tensor([5.0, 5.0, 5.0]))          # old = min(4 - 3, 2) = 1
    assert len(m) == 4 and abs(m.get_mean() - (2.0 * 1 + 5.0 * 3) / 4) < 1e-6
    m.update(torch.arange(10.0))                      # size clipped to 4, old = 0: the batch mean alone
    assert len(m) == 4 and abs(m.get_mean() - 4.5) < 1e-6
