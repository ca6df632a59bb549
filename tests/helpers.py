"""Shared helpers for the parity tests.

Tolerance convention used everywhere: integers / flags / indices are compared
exactly; fp32 values must satisfy |x - ref| <= 1e-5 * max(|ref|, s) where s is
the natural scale of the column (1 for unit-range observations and rewards,
pi for angles, the input's own scale for copied / scaled state).
"""
import json
import os

import numpy as np

from isaacgym_amd import _abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def close(x, ref, scale=1.0, rtol=RTOL):
    """elementwise |x-ref| <= rtol*max(|ref|, scale); NaN == NaN."""
    x = np.asarray(x)
    ref = np.asarray(ref)
    tol = rtol * np.maximum(np.abs(ref), scale)
    ok = np.abs(x.astype(np.float64) - ref.astype(np.float64)) <= tol
    ok |= np.isnan(x) & np.isnan(ref)
    ok |= (x == ref)
    return ok


def assert_close(x, ref, scale=1.0, rtol=RTOL, what=""):
    ok = close(x, ref, scale, rtol)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError("%s: %d/%d outside tolerance; first at %s: got %r want %r" %
                             (what, len(bad), ok.size, i, x[i], ref[i]))


def assert_angle_close(x, ref, rtol=RTOL, what=""):
    """angles: compare modulo 2*pi with scale pi."""
    d = np.abs(np.asarray(x, np.float64) - np.asarray(ref, np.float64))
    d = np.minimum(d, np.abs(d - 2 * np.pi))
    ok = d <= rtol * np.maximum(np.abs(ref), np.pi)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError("%s: %d/%d angles outside tolerance; first at %s: got %r want %r" %
                             (what, len(bad), ok.size, i, x[i], ref[i]))


def bit_equal_fraction(x, ref):
    x = np.ascontiguousarray(x, np.float32).view(np.uint32)
    ref = np.ascontiguousarray(ref, np.float32).view(np.uint32)
    return float((x == ref).mean())


def cartpole_params(meta, seed=0, rng_offset=0):
    return _abi.CartpoleParams(reset_dist=meta["reset_dist"],
                               max_episode_length_m1=float(meta["max_episode_length"] - 1),
                               seed=seed, rng_offset=rng_offset)


def loco_params(meta, g, seed=0, rng_offset=0):
    p = _abi.LocoParams(
        num_dof=meta["num_dof"], num_sensor_floats=meta["num_sensor_floats"], dt=meta["dt"],
        dof_vel_scale=meta["dof_vel_scale"], contact_force_scale=meta["contact_force_scale"],
        angular_velocity_scale=meta["angular_velocity_scale"],
        up_weight=meta["up_weight"], heading_weight=meta["heading_weight"],
        actions_cost_scale=meta["actions_cost_scale"], energy_cost_scale=meta["energy_cost_scale"],
        joints_at_limit_cost_scale=meta["joints_at_limit_cost_scale"],
        termination_height=meta["termination_height"], death_cost=meta["death_cost"],
        max_episode_length_m1=float(meta["max_episode_length"] - 1),
        targets=[1000.0, 0.0, 0.0], inv_start_rot=[-0.0, -0.0, -0.0, 1.0],
        basis_vec0=[1.0, 0.0, 0.0], basis_vec1=[0.0, 0.0, 1.0],
        dof_limits_lower=g["dof_limits_lower"], dof_limits_upper=g["dof_limits_upper"],
        seed=seed, rng_offset=rng_offset)
    if "motor_efforts" in g:
        p.set("motor_efforts", g["motor_efforts"])
        p.max_motor_effort = meta["max_motor_effort"]
    return p


# observation columns that go through atan2 / remainder (library-dependent
# transcendental functions): yaw, roll, angle_to_target
LOCO_ANGLE_COLS = (7, 8, 9)
