"""EpisodeStats: finished-episode reward / length bookkeeping on the device.

What rl_games' agent does on the host after every env step in the reference
(learning/common_agent.py:284-296: `current_rewards += rewards`, `dones.nonzero()`,
`game_rewards.update(current_rewards[done_indices])`, multiply by `1 - dones`; and
the per-index Python loop of utils/rlgames_utils.py:165-174) is one kernel here;
totals accumulate on the device and are read - and summed over the ranks of a
multi-GPU job - only when `totals()` is called (once per horizon is enough).
"""
import ctypes as C

import torch

from . import _abi, _lib, distributed


def sum_records(recs):
    """recs: int64 [world, 4] host tensor of 32-byte records {n_done, sum_reward (f64 bits),
    sum_length (f64 bits), steps} in rank order -> (n_done, sum_reward, sum_length, steps)."""
    n_done = int(recs[:, 0].sum())
    sums = recs[:, 1:3].contiguous().view(torch.float64)
    sr, sl = 0.0, 0.0
    for r in range(recs.shape[0]):  # fixed order: the same value on every rank
        sr += float(sums[r, 0])
        sl += float(sums[r, 1])
    return n_done, sr, sl, int(recs[0, 3])


class EpisodeStats:
    def __init__(self, num_envs, device):
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.current_rewards = torch.zeros(self.num_envs, device=self.device)
        self.current_lengths = torch.zeros(self.num_envs, device=self.device)
        nrec = (self.num_envs + 255) // 256 + 2
        self._ws = torch.zeros(nrec * 4, dtype=torch.int64, device=self.device)  # 32-byte records

    def step(self, rew, done):
        """rew [N] fp32, done [N] int64 (the reset flags `step()` returned); asynchronous."""
        t = dict(rew=rew, done=done, current_rewards=self.current_rewards, current_lengths=self.current_lengths,
                 workspace=self._ws)
        _lib.check_sizes(_abi.EpisodeBuffers, None, t, self.num_envs, "igx_episode_stats_step")
        b, dev = _lib.make_buffers(_abi.EpisodeBuffers, t)
        with torch.cuda.device(dev):
            rc = _lib.lib().igx_episode_stats_step(C.byref(b), C.c_int64(self.num_envs), _lib.current_stream(dev))
        _lib.check(rc)

    def last_step(self):
        """(n_done, sum_reward, sum_length) of the most recent step on this shard (synchronises)."""
        rec = self._ws[4:8].cpu()
        return int(rec[0]), float(rec[1:3].view(torch.float64)[0]), float(rec[1:3].view(torch.float64)[1])

    def totals(self, group=None, reset=True):
        """(n_done, sum_reward, sum_length, steps) since the last reset, summed over all ranks
        in rank order (one 32-byte-per-rank all-gather).  Synchronises."""
        out = sum_records(distributed.gather_records(self._ws[0:4], group).cpu())
        if reset:
            self._ws[0:4].zero_()
        return out

    def means(self, group=None, reset=True):
        """(mean episode reward, mean episode length) of the episodes that finished; None if none did."""
        n, sr, sl, _ = self.totals(group, reset)
        return (sr / n, sl / n) if n else None
