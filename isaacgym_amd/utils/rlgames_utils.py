"""Episode statistics as the learner's observer keeps them, on the device.

Mirror of the reference's `RLGPUAlgoObserver` (utils/rlgames_utils.py:132-211) and of the reward /
length bookkeeping around it in the agent loop (learning/common_agent.py:284-296):

* `process_infos(infos, done_indices)` (:151-183): `infos['episode']` dicts are collected,
  `infos['episode_cumulative'][key]` tensors are accumulated per environment and, for every done
  environment in ascending order, appended to a window of the last `games_to_track` finished
  episodes and zeroed; scalar entries of the (flattened) infos are kept for direct logging.
  The reference walks `done_indices` in a Python loop with `.item()` per index; here the whole
  update is `igx_episode_window_step` (count, scan, apply) and nothing leaves the device.
* `after_print_stats(frame, epoch_num, total_time)` (:185-211): means of the collected
  `episode` infos, mean / min / max of every window, the direct scalars - written through
  `writer.add_scalar(tag, value, step)` exactly as the reference names them.  This is the one
  place that reads device memory (once per epoch).
* `AverageMeter`: the windowed mean rl_games keeps for `game_rewards` / `game_lengths`
  (rl_games.algos_torch.torch_ext.AverageMeter; rl_games is not part of the reference tree, its
  published update rule is restated here).  `GameStats` feeds two of them from the per-step
  records of `EpisodeStats` without a `nonzero()` per step.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from .. import _abi, _lib
from ..episode_stats import EpisodeStats


def _leaves(node, head, separator):
    for key, value in node.items():
        if isinstance(value, (dict, OrderedDict)):
            yield from _leaves(value, head + key + separator, separator)
        else:
            yield head + key, value


def flatten_dict(d, prefix="", separator="."):
    """Nested dicts as one level, keys joined with `separator`, in document order (what utils/utils.py:69-77 returns)."""
    return dict(_leaves(d, prefix, separator))


class AverageMeter:
    """mean over (at most) the last `max_size` values, updated batch-wise:
    mean <- (mean * old + batch_mean * size) / (old + size) with size clipped to max_size and
    old = min(max_size - size, current_size)."""

    def __init__(self, max_size):
        self.max_size = int(max_size)
        self.current_size = 0
        self.mean = 0.0

    def update_batch(self, count, batch_sum):
        """`count` values whose sum is `batch_sum` (what update(values) sees as values.mean(), len)."""
        count = int(count)
        if count == 0:
            return
        new_mean = float(np.float32(batch_sum / count))
        size = int(np.clip(count, 0, self.max_size))
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = float(np.float32((np.float32(self.mean) * np.float32(old_size) + np.float32(new_mean) * np.float32(size))
                                     / np.float32(size_sum)))

    def update(self, values):
        values = torch.as_tensor(values).float().reshape(-1)
        self.update_batch(values.numel(), float(values.double().sum()) if values.numel() else 0.0)

    def clear(self):
        self.current_size = 0
        self.mean = 0.0

    def __len__(self):
        return self.current_size

    def get_mean(self):
        return self.mean


class GameStats:
    """game_rewards / game_lengths of the agent loop (learning/common_agent.py:284-296) without the
    per-step `dones.nonzero()`: `step(rew, dones)` is the fused bookkeeping kernel; the per-step
    {n_done, sum_reward, sum_length} records wait in a device ring and `flush()` replays them, in step
    order, into the two windowed meters (one read of device memory per horizon)."""

    def __init__(self, num_envs, device, games_to_track=100, horizon=64):
        self.stats = EpisodeStats(num_envs, device)
        self.game_rewards = AverageMeter(games_to_track)
        self.game_lengths = AverageMeter(games_to_track)
        self._ring = torch.zeros(int(horizon), 4, dtype=torch.int64, device=self.stats.device)
        self._pending = 0

    def step(self, rew, dones):
        if self._pending == self._ring.shape[0]:
            self.flush()
        self.stats.step(rew, dones)
        self._ring[self._pending].copy_(self.stats._ws[4:8])  # this step's record, device to device
        self._pending += 1

    def flush(self):
        if not self._pending:
            return
        recs = self._ring[:self._pending].cpu()
        sums = recs[:, 1:3].contiguous().view(torch.float64)
        for i in range(self._pending):
            n = int(recs[i, 0])
            self.game_rewards.update_batch(n, float(sums[i, 0]))
            self.game_lengths.update_batch(n, float(sums[i, 1]))
        self._pending = 0


class RLGPUAlgoObserver:
    """Allows us to log stats from the env along with the algorithm running stats (the reference's
    class of the same name).  `algo` needs `games_to_track`, `writer` and `device`."""

    def __init__(self):
        self.algo = None
        self.writer = None
        self.ep_infos = []
        self.direct_info = {}
        self.episode_cumulative = dict()      # key -> [N] accumulator (device)
        self.new_finished_episodes = False
        self._keys = []
        self._window = None                   # [num_keys, G] rings
        self._ws = None
        self._seen_total = 0

    def before_init(self, base_name, config, experiment_name):  # rl_games AlgoObserver hooks this class leaves empty
        pass

    def after_steps(self):
        pass

    def after_clear_stats(self):
        pass

    def after_init(self, algo):
        self.algo = algo
        self.writer = self.algo.writer

    # -- the reference signature: indices from dones.nonzero(); kept for drop-in callers
    def process_infos(self, infos, done_indices):
        assert isinstance(infos, dict), "RLGPUAlgoObserver expects dict info"
        dones = None
        if "episode_cumulative" in infos and len(infos["episode_cumulative"]):
            first = next(iter(infos["episode_cumulative"].values()))
            dones = torch.zeros(first.shape[0], dtype=torch.int64, device=first.device)
            idx = torch.as_tensor(done_indices, device=first.device).reshape(-1).long()
            dones[idx] = 1
        self.process_infos_flags(infos, dones)

    # -- the same update from the done FLAGS step() returned: no nonzero() anywhere
    def process_infos_flags(self, infos, dones):
        assert isinstance(infos, dict), "RLGPUAlgoObserver expects dict info"
        if "episode" in infos:
            self.ep_infos.append(infos["episode"])
        if "episode_cumulative" in infos and len(infos["episode_cumulative"]):
            self._window_step(infos["episode_cumulative"], dones)
        if len(infos) > 0:  # allow direct logging from env
            infos_flat = flatten_dict(infos, prefix="", separator="/")
            self.direct_info = {}
            for k, v in infos_flat.items():
                if isinstance(v, float) or isinstance(v, int) or (isinstance(v, torch.Tensor) and len(v.shape) == 0):
                    self.direct_info[k] = v

    def _window_step(self, cumulative, dones):
        keys = list(cumulative.keys())
        first = cumulative[keys[0]]
        n, dev = first.shape[0], first.device
        G = int(self.algo.games_to_track)
        if not self._keys:
            if len(keys) > _abi.IGX_EPISODE_WINDOW_MAX_KEYS:
                raise ValueError("at most %d episode_cumulative keys" % _abi.IGX_EPISODE_WINDOW_MAX_KEYS)
            self._keys = keys
            for k in keys:
                self.episode_cumulative[k] = torch.zeros(n, dtype=torch.float32, device=dev)
            self._window = torch.zeros(len(keys), G, dtype=torch.float32, device=dev)
            self._ws = torch.zeros(_abi.IGX_EPISODE_WINDOW_WORKSPACE_BYTES(n) // 8, dtype=torch.int64, device=dev)
        elif keys != self._keys:
            raise ValueError("the keys of infos['episode_cumulative'] changed: %s -> %s" % (self._keys, keys))
        if dones is None or dones.shape[0] != n:
            raise ValueError("dones must hold one flag per environment")
        vals = []
        for k in keys:
            v = cumulative[k]
            if v.dtype != torch.float32 or not v.is_contiguous():
                v = v.float().contiguous()
            if v.shape[0] != n or v.dim() != 1:
                raise ValueError("infos['episode_cumulative'][%r] must be a [num_envs] tensor" % k)
            vals.append(v)
        w = _abi.EpisodeWindow(done=_lib.tensor_ptr(dones, torch.int64), num_keys=len(keys), games_to_track=G,
                               window=_lib.tensor_ptr(self._window, torch.float32), workspace=_lib.tensor_ptr(self._ws))
        w.set("values", [_lib.tensor_ptr(v, torch.float32) for v in vals])
        w.set("cumulative", [_lib.tensor_ptr(self.episode_cumulative[k], torch.float32) for k in keys])
        with torch.cuda.device(dev):
            rc = _lib.lib().igx_episode_window_step(C.byref(w), C.c_int64(n), _lib.current_stream(dev))
        _lib.check(rc)
        self._keep = vals  # the launch is asynchronous: converted copies stay alive until the next call

    def windows(self):
        """({key: numpy array of the window's values, oldest first}, episodes finished so far); synchronises."""
        if not self._keys:
            return {}, 0
        G = self._window.shape[1]
        total = int(self._ws[0])
        valid = min(total, G)
        ring = self._window.cpu().numpy()
        order = [(i % G) for i in range(total - valid, total)]
        return {k: ring[j, order].astype(np.float64) for j, k in enumerate(self._keys)}, total

    def after_print_stats(self, frame, epoch_num, total_time):
        if self.ep_infos:  # :186-198: every key of the first dict, mean over everything collected since the last call
            dev = self.algo.device
            for key in self.ep_infos[0]:
                parts = [torch.tensor([], device=dev)]
                for ep_info in self.ep_infos:
                    v = ep_info[key]
                    v = v if isinstance(v, torch.Tensor) else torch.Tensor([v])  # plain numbers
                    parts.append((v.unsqueeze(0) if v.dim() == 0 else v).to(dev))  # and 0-dim tensors
                self.writer.add_scalar("Episode/" + key, torch.mean(torch.cat(parts)), epoch_num)
            self.ep_infos.clear()

        if self._keys:  # :200-207: the windowed rows appear only in epochs that finished at least one episode
            win, total = self.windows()
            finished_more = total > self._seen_total
            self._seen_total = total
            self.new_finished_episodes = False
            if finished_more:
                for key in self._keys:
                    vals = win[key]
                    for tag, stat in (("episode_cumulative/%s" % key, np.mean(vals)),
                                      ("episode_cumulative_min/%s_min" % key, np.min(vals)),
                                      ("episode_cumulative_max/%s_max" % key, np.max(vals))):
                        self.writer.add_scalar(tag, stat, frame)

        for name, value in self.direct_info.items():  # :209-211: each direct scalar against three x axes
            for axis, x in (("frame", frame), ("iter", epoch_num), ("time", total_time)):
                self.writer.add_scalar("%s/%s" % (name, axis), value, x)
