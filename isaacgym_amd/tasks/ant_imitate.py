"""AntImitate: the fork's pose-imitation variant of Ant (reference tasks/ant_imitate.py).
Observations as Ant; the reward is exp(-|joint pose - target frame|_1) with a reset
threshold at max(1.8, 0.99-quantile over ALL environments) (:447-555), evaluated in
fp64 like the reference (its motion table comes from np.loadtxt).  One fused kernel, an
exact device quantile and one finishing kernel per step.  In an environment-sharded job the quantile
is the one exchange: by default two collectives per step (ops.WindowQuantile: a histogram over a window
of the key space that follows the value, then the keys of one bin), or the six all-reduced histograms
of the radix select (cfg env.quantileCollectives: "radix"), which is also where data goes that the
window cannot resolve."""
import numpy as np
import torch

from .. import _abi, distributed, ops
from .locomotion import Ant


class AntImitate(Ant):
    QUANTILE_THRESHOLD = 0.99
    DISTANCE_FLOOR = 1.8

    def __init__(self, cfg, rl_device, sim_device, graphics_device_id=0, headless=True, virtual_screen_capture=False,
                 force_render=False, sim=None, seed=0, motion=None, process_group=None):
        super().__init__(cfg, rl_device, sim_device, graphics_device_id, headless, virtual_screen_capture,
                         force_render, sim=sim, seed=seed)
        if motion is None:
            motion = np.loadtxt(cfg["motionToImitate"])  # ant_imitate.py:87-90
        motion = np.ascontiguousarray(motion, dtype=np.float64)
        if motion.ndim != 2 or motion.shape[1] != self.num_dof:
            raise ValueError("the motion to imitate must be [frames, %d]" % self.num_dof)
        self.motion_to_imitate = torch.from_numpy(motion).to(self.device)
        self.target_pose_idx = torch.ones(self.num_envs, device=self.device, dtype=torch.long)
        self._pose_diff = torch.zeros(self.num_envs, device=self.device, dtype=torch.float64)
        self._qws = torch.empty(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=self.device)
        self._group = process_group
        self._im = None
        self._n_total = None  # environments of all ranks (asked once)
        self._window = None
        self.quantile_collectives = self.cfg["env"].get("quantileCollectives", "window")
        if self.quantile_collectives not in ("window", "radix"):
            raise ValueError("env.quantileCollectives must be 'window' or 'radix'")

    def _imitate_args(self, n_total):
        key = (n_total, float(self.death_cost))
        if self._im is None or self._im[0] != key:  # the tensors it points at live as long as the task
            self._im = (key, ops.imitate_args(self.motion_to_imitate, self.target_pose_idx, self._pose_diff, self._qws,
                                              self.QUANTILE_THRESHOLD, self.DISTANCE_FLOOR, float(self.death_cost), n_total))
        return self._im[1]

    def post_physics_step(self):
        # ant_imitate.py reset_idx re-randomises like Ant's, i.e. on steps where some env is reset
        if self.randomize and bool(self.reset_buf.any()):
            self.apply_randomizations(self.randomization_params)
        world = distributed.world_size(self._group)
        p, b = self._params(), self._buffers()
        if self._can_fuse_tail():  # the clamped copy (observe kernel) and the time-out flags (finish kernel)
            self._fuse_tail(p, b)
        if world == 1:
            ops.ant_imitate_step(p, b, self._imitate_args(self.num_envs), self.num_envs, self._call_cache)
        else:
            if self._n_total is None:
                counts = distributed.gather_records(torch.tensor([self.num_envs], device=self.device), self._group)
                self._n_total = int(counts.sum())
            im = self._imitate_args(self._n_total)
            ops.ant_imitate_observe(p, b, im, self.num_envs, self._call_cache)
            done = False
            if self.quantile_collectives == "window" and self._n_total < (1 << 31):
                if self._window is None:
                    self._window = ops.WindowQuantile(self.device)
                # the value goes where igx_ant_imitate_finish reads it: the result field of the select's workspace
                done = self._window.run(self._pose_diff, self.QUANTILE_THRESHOLD, self._n_total,
                                        lambda h: distributed.all_reduce_sum(h, self._group),
                                        lambda t: distributed.gather_records(t, self._group),
                                        out=self._qws.view(torch.float64)[5:6])
            if not done:
                ops.quantile_sharded(self._pose_diff, self.QUANTILE_THRESHOLD, self._n_total, self._qws,
                                     lambda h: distributed.all_reduce_sum(h, self._group))
            ops.ant_imitate_finish(p, b, im, self.num_envs, self._call_cache)
        self.extras["true_objective"] = self._true_objective

    @property
    def pose_threshold(self):
        """max(1.8, quantile) of the last step (synchronises)."""
        return max(self.DISTANCE_FLOOR, float(self._qws.view(torch.float64)[5]))
