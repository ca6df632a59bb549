"""Time the REFERENCE's own torch.jit functions on this container's CPU cores (build container
only: needs /root/reference).  One step = progress += 1, compute_*_observations,
compute_*_reward and the in-place buffer copies the task class does (ant.py:299-330,
humanoid.py:322-357, cartpole.py:135-146, :185-201).  Prints one JSON line per task."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgym_amd import synthetic  # noqa: E402
from oracle import ref_shim  # noqa: E402


def timed(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def loco(name, spec, n):
    mod = ref_shim.load("isaacgymenvs.tasks." + name)
    t = synthetic.loco_state(spec, n, 0, "cpu")
    D = spec["num_dof"]
    lo, hi = synthetic.loco_limits(D)
    lo, hi = torch.tensor(lo), torch.tensor(hi)
    dof_pos, dof_vel = t["dof_state"][..., 0], t["dof_state"][..., 1]
    targets = torch.tensor([1000.0, 0, 0]).repeat(n, 1)
    inv = torch.tensor([0.0, 0, 0, 1]).repeat(n, 1)
    b0, b1 = torch.tensor([1.0, 0, 0]).repeat(n, 1), torch.tensor([0.0, 0, 1]).repeat(n, 1)
    obs_buf, rew_buf, reset_buf = t["obs"], t["rew"], t["reset"]
    pot, prev = t["potentials"], t["prev_potentials"]
    up, head = t["up_vec"], t["heading_vec"]
    if name == "ant":
        def step():
            t["progress"].add_(1)
            o = mod.compute_ant_observations(obs_buf, t["root_states"], targets, pot, inv, dof_pos, dof_vel, lo, hi,
                                             spec["dof_vel_scale"], t["sensors"], t["actions"], spec["dt"],
                                             spec["contact_force_scale"], b0, b1, 2)
            obs_buf[:], pot[:], prev[:], up[:], head[:] = o
            r = mod.compute_ant_reward(obs_buf, reset_buf, t["progress"], t["actions"], spec["up_weight"],
                                       spec["heading_weight"], pot, prev, spec["actions_cost_scale"],
                                       spec["energy_cost_scale"], spec["joints_at_limit_cost_scale"],
                                       spec["termination_height"], spec["death_cost"], float(spec["max_episode_length"]))
            rew_buf[:], reset_buf[:] = r
    else:
        p = synthetic.loco_params(spec)
        efforts = torch.tensor(list(p.motor_efforts)[:D])
        def step():
            t["progress"].add_(1)
            o = mod.compute_humanoid_observations(obs_buf, t["root_states"], targets, pot, inv, dof_pos, dof_vel,
                                                  t["dof_force"], lo, hi, spec["dof_vel_scale"], t["sensors"],
                                                  t["actions"], spec["dt"], spec["contact_force_scale"],
                                                  spec["angular_velocity_scale"], b0, b1)
            obs_buf[:], pot[:], prev[:], up[:], head[:] = o
            r = mod.compute_humanoid_reward(obs_buf, reset_buf, t["progress"], t["actions"], spec["up_weight"],
                                            spec["heading_weight"], pot, prev, spec["actions_cost_scale"],
                                            spec["energy_cost_scale"], spec["joints_at_limit_cost_scale"],
                                            float(p.max_motor_effort), efforts, spec["termination_height"],
                                            spec["death_cost"], float(spec["max_episode_length"]))
            rew_buf[:], reset_buf[:] = r
    return step


def main():
    ref_shim.install()
    threads = int(os.environ.get("IGX_REF_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    n = int(os.environ.get("IGX_REF_ENVS", 1 << 20))
    for name, spec in (("ant", synthetic.ANT), ("humanoid", synthetic.HUMANOID)):
        med, best = timed(loco(name, spec, n))
        print(json.dumps({"task": name, "envs": n, "threads": threads, "torch": torch.__version__,
                          "median_s": med, "min_s": best, "env_steps_per_s_median": n / med}), flush=True)


if __name__ == "__main__":
    main()
