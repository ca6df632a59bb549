"""Random disturbance forces on the manipulated object, shared by AllegroHand and ShadowHand.

The reference keeps `rb_forces [N, num_bodies, 3]` and `random_force_prob [N]` on the task
(tasks/allegro_hand.py:191-196); `reset_idx` zeroes the forces of a reset environment and re-draws its
probability log-uniformly in `forceProbRange` (:532, :558-559); with `forceScale > 0` every
`pre_physics_step` decays all forces, picks environments by a Bernoulli draw against their probability and
writes `randn * mass * forceScale` into the object's body rows (:617-625) before handing the tensor to
`gym.apply_rigid_body_force_tensors`.  All of that is one kernel here (`igx_hand_random_forces`), launched
ahead of the task's pre-physics kernel, with no `nonzero()`.

With `forceScale == 0` (the default of both task configs) the kernel is not launched: the forces stay zero
and nothing reads `random_force_prob`.
"""
import torch

from .. import ops


class HandForcesMixin:
    def _init_forces(self, env):
        """Call after VecTask.__init__ (needs device, num_envs, dt, sim)."""
        self.force_prob_range = env.get("forceProbRange", [0.001, 0.1])
        self.force_decay = env.get("forceDecay", 0.99)
        self.force_decay_interval = env.get("forceDecayInterval", 0.08)
        asset = self.sim.asset
        self.num_bodies = int(asset.get("num_bodies", getattr(self, "num_bodies", 0) or 1))
        self.object_rb_handles = [int(asset.get("object_rb_handle", self.num_bodies - 1))]
        self.object_rb_masses = [float(asset.get("object_rb_mass", 1.0))]
        dev, n = self.device, self.num_envs
        self.rb_forces = torch.zeros((n, self.num_bodies, 3), dtype=torch.float, device=dev)
        decay, lo, hi = ops.hand_force_constants(self.force_decay, self.dt, self.force_decay_interval, self.force_prob_range)
        self._force_consts = dict(decay=decay, log_prob_lo=lo, log_prob_hi=hi)
        # :193-194: the first probabilities are drawn like the re-draws on reset
        gen = torch.Generator(device="cpu").manual_seed(int(getattr(self, "seed", 0)) + 7919)
        self.random_force_prob = torch.exp((lo - hi) * torch.rand(n, generator=gen) + hi).to(dev)
        self.force_rand = None   # optional [N,5] pre-drawn (u_prob, u_sel, z0, z1, z2) per environment
        self._force_draws = 0

    def _apply_random_forces(self):
        """The rb_forces part of pre_physics_step; must run before the pre-physics kernel clears reset_buf."""
        if not self.force_scale > 0.0:
            return
        self._force_draws += 1
        ops.hand_random_forces(self.rb_forces, self.random_force_prob, self.reset_buf, self.object_rb_handles[0],
                               self.object_rb_masses[0], self.force_scale, rand=self.force_rand,
                               seed=(int(self.seed) * 0x9E3779B1 + 0x51ED) & 0xFFFFFFFFFFFFFFFF,
                               rng_offset=self._force_draws, rng_offset_dev=self._force_rng_dev(), env_offset=self.env_offset,
                               **self._force_consts)
        self.sim.apply_rigid_body_forces(self.rb_forces)

    def _force_rng_dev(self):
        return self._rng_dev()  # the device-resident draw counter of a replayed graph, or None
