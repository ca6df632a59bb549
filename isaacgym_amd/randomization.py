"""Observation / action noise of the domain-randomisation block: the host side.

Mirror of the non-physical part of VecTask.apply_randomizations
(tasks/base/vec_task.py:630-729): the schedule arithmetic that turns a
`randomization_params` entry into the parameters of a noise lambda, in the same
Python-float (double) operations and order, so that the parameters - and with
them the fp32 constants the kernels receive - are the reference's bit for bit.
The noise itself is applied by igx_noise_apply / igx_action_stage_step /
igx_vec_task_tail_step.  Randomisation of simulator properties (sim_params,
actor_params) needs a physics engine and has no counterpart here.
"""

NONPHYSICAL = ("observations", "actions")


def schedule_scaling(cfg, last_step):
    """:663-673"""
    sched_type = cfg["schedule"] if "schedule" in cfg else None
    sched_step = cfg["schedule_steps"] if "schedule" in cfg else None
    if sched_type == "linear":
        return 1.0 / sched_step * min(last_step, sched_step)
    if sched_type == "constant":
        return 0 if last_step < sched_step else 1
    return 1


def noise_params(cfg, last_step):
    """One entry of self.dr_randomizations without its lambda (:675-729): the reference's
    keys ('mu', 'var', 'mu_corr', 'var_corr' or 'lo', 'hi', 'lo_corr', 'hi_corr') plus
    'operation' and 'distribution'."""
    dist, op_type = cfg["distribution"], cfg["operation"]
    s = schedule_scaling(cfg, last_step)
    if dist == "gaussian":
        mu, var = cfg["range"]
        mu_corr, var_corr = cfg.get("range_correlated", [0., 0.])
        if op_type == "additive":
            mu *= s
            var *= s
            mu_corr *= s
            var_corr *= s
        elif op_type == "scaling":
            var = var * s
            mu = mu * s + 1.0 * (1.0 - s)
            var_corr = var_corr * s
            mu_corr = mu_corr * s + 1.0 * (1.0 - s)
        out = {"mu": mu, "var": var, "mu_corr": mu_corr, "var_corr": var_corr}
    elif dist == "uniform":
        lo, hi = cfg["range"]
        lo_corr, hi_corr = cfg.get("range_correlated", [0., 0.])
        if op_type == "additive":
            lo *= s
            hi *= s
            lo_corr *= s
            hi_corr *= s
        elif op_type == "scaling":
            lo = lo * s + 1.0 * (1.0 - s)
            hi = hi * s + 1.0 * (1.0 - s)
            lo_corr = lo_corr * s + 1.0 * (1.0 - s)
            hi_corr = hi_corr * s + 1.0 * (1.0 - s)
        out = {"lo": lo, "hi": hi, "lo_corr": lo_corr, "hi_corr": hi_corr}
    else:
        raise ValueError("noise distribution must be 'gaussian' or 'uniform', got %r" % (dist,))
    out["distribution"] = dist
    # operator.add if op_type == 'additive' else operator.mul (:665)
    out["operation"] = "additive" if op_type == "additive" else "scaling"
    return out


def bind_operations(entries):
    """Both lambdas close over the one loop variable `op` (:665, :700, :727), which they read
    when they are called - after the loop: every lambda made by one apply_randomizations call
    applies the operation of the LAST parameter set that call processed ('actions' when both
    are randomised), whatever its own 'operation' says.  `entries`: the entries made by one
    call, in processing order."""
    if entries:
        for e in entries:
            e["operation"] = entries[-1]["operation"]


def noise_coefficients(p):
    """(corr_scale, corr_shift, scale, shift) of noise = ((corr * corr_scale + corr_shift) +
    fresh * scale) + shift (:699-701, :726-727); the differences are taken in double here and
    rounded to fp32 when they meet the fp32 tensors, as the reference's are."""
    if "mu" in p:
        return p["var_corr"], p["mu_corr"], p["var"], p["mu"]
    return p["hi_corr"] - p["lo_corr"], p["lo_corr"], p["hi"] - p["lo"], p["lo"]
