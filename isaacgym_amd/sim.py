"""Stand-ins for the physics engine behind the task classes.

The reference's tasks wrap PhysX-owned device memory (`gym.acquire_*_tensor` +
`gymtorch.wrap_tensor`) and call `gym.simulate`.  The closed-source simulator
is not part of this package; a task here is handed a *sim* object that owns
the same state tensors (same names, shapes and AoS layouts) and advances them
in `simulate()`.  Two are provided:

  SyntheticSim  seeded random states (benchmarks, demos, tests);
  ReplaySim     recorded frames copied into the tensors, one per simulate().

A binding to a real engine implements the same five members: `tensors`,
`asset`, `simulate()`, `refresh()`, `notify_*()`.
"""
import torch

from . import _abi, synthetic


class TensorSim:
    def __init__(self, num_envs, device):
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.tensors = {}   # name -> state tensor owned by the simulator
        self.asset = {}     # per-asset constants (dof limits, motor efforts, ...)
        self.num_simulate_calls = 0

    def acquire(self, name):
        if name not in self.tensors:
            raise KeyError("this sim provides no %r tensor (has: %s)" % (name, sorted(self.tensors)))
        return self.tensors[name]

    def simulate(self):
        self.num_simulate_calls += 1

    def refresh(self):
        """gym.refresh_*_tensor: the tensors are shared memory here, nothing to do."""

    # the tasks write resets straight into the shared tensors; these are the
    # notifications gym.set_*_tensor_indexed would carry to the engine
    def notify_dof_state(self, env_ids=None):
        pass

    def notify_root_state(self, rows, actor_ids=None):
        pass

    def apply_dof_forces(self, forces):
        self.tensors["dof_actuation_force"] = forces

    def apply_dof_position_targets(self, targets):
        self.tensors["dof_position_target"] = targets

    def apply_rigid_body_forces(self, forces):
        """gym.apply_rigid_body_force_tensors(sim, forces, None, LOCAL_SPACE)"""
        self.tensors["rigid_body_force"] = forces


_PHYSICS_OWNED = {
    "Cartpole": ("dof_state",),
    "CartpoleSample": ("dof_state",),
    "Ant": ("root_states", "dof_state", "sensors"),
    "AntImitate": ("root_states", "dof_state", "sensors"),
    "Humanoid": ("root_states", "dof_state", "sensors", "dof_force"),
    "HumanoidImitate": ("root_states", "dof_state", "rigid_body_states"),
    "AllegroHand": ("dof_state", "dof_force", "root_state_tensor"),
    "HumanoidAMP": ("root_states", "dof_state", "rigid_body_state", "contact_forces"),
    "ShadowHand": ("dof_state", "dof_force", "vec_sensor_tensor", "rigid_body_states", "root_state_tensor"),
}


def _draw(task, n, seed, device, **kw):
    if task in ("Cartpole", "CartpoleSample"):
        return synthetic.cartpole_state(n, seed, device)
    if task in ("Ant", "AntImitate"):
        return synthetic.loco_state(synthetic.ANT, n, seed, device)
    if task == "Humanoid":
        return synthetic.loco_state(synthetic.HUMANOID, n, seed, device)
    if task == "HumanoidImitate":
        return synthetic.him_state(n, seed, device)
    if task == "AllegroHand":
        return synthetic.allegro_state(n, seed, device, kw.get("obs_type", "full_state"))
    if task == "ShadowHand":
        return synthetic.shadow_state(n, seed, device, kw.get("obs_type", "full_state"), kw.get("asymmetric", False))
    if task == "HumanoidAMP":
        return synthetic.amp_state(n, seed, device, kw.get("num_amp_obs_steps", 2))
    raise KeyError(task)


class SyntheticSim(TensorSim):
    """Seeded random 'physics': simulate() swaps in the next of `num_sets`
    pre-drawn states (device-to-device copies, no host work)."""

    def __init__(self, task, num_envs, device, seed=0, num_sets=2, **kw):
        super().__init__(num_envs, device)
        self.task = task
        first = _draw(task, num_envs, seed, self.device, **kw)
        self.tensors = dict(first)
        owned = _PHYSICS_OWNED[task]
        owned = [k for k in owned if first.get(k) is not None]
        self._sets = [{k: first[k].clone() for k in owned}]
        for i in range(1, num_sets):
            nxt = _draw(task, num_envs, seed + 1000 * i, self.device, **kw)
            self._sets.append({k: nxt[k] for k in owned})
        if task in ("Ant", "AntImitate", "Humanoid"):
            spec = synthetic.HUMANOID if task == "Humanoid" else synthetic.ANT
            lo, hi = synthetic.loco_limits(spec["num_dof"])
            p = synthetic.loco_params(spec)
            self.asset = dict(dof_limits_lower=lo, dof_limits_upper=hi,
                              motor_efforts=list(p.motor_efforts)[:spec["num_dof"]], num_dof=spec["num_dof"])
        elif task == "HumanoidImitate":
            lo, hi = synthetic.him_limits()
            self.asset = dict(dof_limits_lower=lo, dof_limits_upper=hi, num_dof=_abi.IGX_HIM_DOF)
        elif task == "ShadowHand":
            p = synthetic.shadow_params()
            self.asset = dict(dof_limits_lower=list(p.dof_lower), dof_limits_upper=list(p.dof_upper),
                              num_dof=_abi.IGX_SHADOW_DOF, actuated_dof_indices=list(synthetic.SHADOW_ACTUATED),
                              fingertip_handles=list(synthetic.SHADOW_FINGERTIPS),
                              num_bodies=int(p.num_bodies), object_rb_handle=int(p.num_bodies) - 2, object_rb_mass=0.1)
        elif task == "AllegroHand":
            p = synthetic.allegro_params()
            self.asset = dict(dof_limits_lower=list(p.dof_lower), dof_limits_upper=list(p.dof_upper),
                              num_dof=_abi.IGX_ALLEGRO_DOF, num_bodies=23, object_rb_handle=21, object_rb_mass=0.1)

    def simulate(self):
        super().simulate()
        s = self._sets[self.num_simulate_calls % len(self._sets)]
        for k, v in s.items():
            self.tensors[k].copy_(v)


class ReplaySim(TensorSim):
    """Plays recorded simulator states: frame t (name -> array) is copied into
    the shared tensors by the t-th simulate() call."""

    def __init__(self, num_envs, device, initial, frames, asset=None):
        super().__init__(num_envs, device)
        self.tensors = {k: torch.as_tensor(v).to(self.device).contiguous().clone() for k, v in initial.items()}
        self.frames = frames
        self.asset = dict(asset or {})

    def simulate(self):
        f = self.frames[self.num_simulate_calls]
        for k, v in f.items():
            self.tensors[k].copy_(torch.as_tensor(v).reshape(self.tensors[k].shape))
        super().simulate()
