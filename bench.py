#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused Ant post_physics_step
(observations + reward + reset bookkeeping) at 1,048,576 envs per GPU.

    python bench.py --gpus N --steps K --warmup W

For N > 1 there is one rank per GPU: either torch.distributed.run started them
(WORLD_SIZE is set), or this process starts N fresh children itself before it
touches the GPU, waits for them and relays rank 0's JSON line.  Environments
shard into independent contiguous blocks (weak scaling: 2^20 envs per GPU), so
the step path has no collective - RCCL is used for the barrier and for the
MAX-over-ranks of the timed region only.

With `--backend nccl` (the default) and more than one rank, an RCCL bring-up that
fails on any rank ends every rank with a non-zero exit code; `--allow-gloo-fallback`
lets the barrier and the MAX run over gloo instead, and the JSON names the backend
of every collective that was used (`collectives`, `rccl_ranks`).

Before the W warm-up steps the run's own buffers take `--settle` untimed launches
(100 by default, `config.settle_launches` in the JSON): a process's first ~75
launches run 2-9 % slower than all the ones after them (tools/step_trend.py), and
the steady state is what a training job lives in.  The K timed steps are exactly K.

Rank 0 prints ONE JSON line.  `value` is whole-job env-steps/s with all inputs
resident in HBM.  After the headline measurement the other configurations of the
path (Humanoid, AllegroHand post / pre, HumanoidAMP step, motion-library samples)
are timed for ~50 launches each and reported under `configs`.  `roofline` prices the dominant (only) kernel against the HBM
peak using 556 algorithmic bytes per env-step; `cpu_baseline` times the CPU
checker (oracle/, a scalar C restatement of the reference's torch.jit path,
OpenMP over envs) and the torch-op restatement of the reference's CPU pipeline
(oracle/torch_ref.py) on this box's host cores for a bounded sample each.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "env-steps/sec obs+reward+reset (Ant, 1M envs) at 1/2/4/8 GPUs; % HBM roofline"
ENVS_PER_GPU = 1 << 20
ANT_BYTES_PER_ENV_STEP = 556      # read 264 + write 292 (obs/reward/reset path of one Ant env)
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak
NUM_INPUT_SETS = 4                # rotate so no step re-reads lines still in the 256 MiB Infinity Cache


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed launches on the run's own buffers before the warm-up steps: the first ~75 launches "
                         "of a process run 2-9 %% slower than the ones after them (tools/step_trend.py)")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--task", default="ant", choices=["ant", "humanoid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="with --backend nccl: if RCCL does not come up on every rank, run the barrier / MAX over gloo "
                         "instead of failing (the JSON then says so)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other configurations (configs in the JSON)")
    ap.add_argument("--config-steps", type=int, default=50, help="timed launches per extra configuration")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this device index instead of LOCAL_RANK")
    ap.add_argument("--force-group", action="store_true",
                    help="rehearsal only: join a process group even as a single rank, so that the barrier and the MAX "
                         "of the N > 1 path run through the backend on a one-GPU box")
    ap.add_argument("--episode-stats", action="store_true",
                    help="also run the episode-statistics kernel every step (inside the timed region) and "
                         "all-gather the totals over the ranks afterwards")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline sample size in steps (0 = auto)")
    ap.add_argument("--spawn-timeout", type=float, default=900.0,
                    help="seconds the self-started ranks of an N > 1 run may take")
    return ap.parse_args()


def _time_loop(fn, budget_s, max_steps):
    fn(0)  # warm-up (page faults, thread pool start)
    t0 = time.perf_counter()
    steps = 0
    while True:
        fn(steps + 1)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or steps >= max_steps:
            return steps, dt


def cpu_baseline(spec, n, task, budget_s=10.0):
    """The same workload on this box's host cores, on a bounded sample (about `budget_s` seconds each):
    `torch_ref` = the step as the reference runs it on a CPU pipeline - unfused torch tensor ops, one ATen
    kernel per arithmetic step (oracle/torch_ref.py, bit-checked against the recording of the reference's
    Ant class) - and `port` = the scalar C restatement with OpenMP over environments (oracle/igx_oracle.c).
    Both use the threads this process was given (OMP_NUM_THREADS / torch's default); host_cpus is what the
    machine has."""
    import torch
    from isaacgym_amd import synthetic
    from oracle import oracle

    host_cpus = os.cpu_count() or 1
    env_threads = os.environ.get("OMP_NUM_THREADS")
    threads = int(env_threads) if env_threads and env_threads.isdigit() else host_cpus
    out = {"unit": "env-steps/s", "host_cpus": host_cpus}

    cores = oracle.set_num_threads(threads)
    p = synthetic.loco_params(spec, seed=1, rng_offset=0)
    t = synthetic.loco_state(spec, n, 0, "cpu")
    bufs = {k: v.numpy() for k, v in t.items()}
    step = oracle.ant_step if task == "ant" else oracle.humanoid_step

    def port_step(i):
        p.rng_offset = i
        step(p, bufs, n)
    steps, dt = _time_loop(port_step, budget_s, 4096)
    out["port"] = {"value": n * steps / dt, "cores": cores,
                   "sample": "%d steps of %d envs (%s), %.1f s wall, C restatement + OpenMP" % (steps, n, task, dt)}

    if task == "ant":
        from oracle import torch_ref
        tthreads = torch_ref.configure_like_the_task(threads)
        tb = synthetic.loco_state(spec, n, 0, "cpu")
        tb["true_objective"] = torch.zeros(n)
        ref = torch_ref.AntStepRef(synthetic.loco_params(spec, seed=1), tb, n)
        with torch.no_grad():
            steps, dt = _time_loop(lambda i: ref.step(), budget_s, 4096)
        out["torch_ref"] = {"value": n * steps / dt, "cores": tthreads,
                            "sample": "%d steps of %d envs (%s), %.1f s wall, unfused torch CPU ops (torch %s)"
                                      % (steps, n, task, dt, torch.__version__)}
        head, kind = out["torch_ref"], "port"
        out["headline_entry"] = "torch_ref"
    else:
        head, kind = out["port"], "port"
        out["headline_entry"] = "port"
    # kind: neither entry is the reference's own code (it cannot travel to this box); both are this repository's
    # restatements ("port").  `headline_entry` names the one value / cores / sample were taken from: the unfused
    # torch-op pipeline is how the reference executes the step, the C + OpenMP restatement is the stronger CPU figure.
    out.update(value=head["value"], cores=head["cores"], sample=head["sample"], kind=kind)
    return out


TRAFFIC_FILE = "profiles/traffic_ant.json"


def load_traffic():
    """HBM bytes per launch from the committed PMC summary, if there is one: a constant read from a file of an
    earlier rocprofv3 --pmc run of this same command, NOT a quantity of this run (the JSON says so)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            d = json.load(f)
        return float(d["hbm_bytes_per_launch"])
    except Exception:
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (rank -> device as the reference's rlgames_utils.py:91-112), relay rank 0's output."""
    from isaacgym_amd.launch import spawn_ranks
    rc, out = spawn_ranks([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                          timeout=args.spawn_timeout)
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


class Group:
    """The process groups of a run: `control` (gloo, host tensors: agreement between the ranks) and, when RCCL came
    up on every rank, `device` (nccl = RCCL, device tensors).  `backend` is what the timed barrier / MAX use."""

    def __init__(self, control, device, backend, rccl_ranks, note=None):
        self.control, self.device, self.backend, self.rccl_ranks, self.note = control, device, backend, rccl_ranks, note

    @property
    def data(self):
        return self.device if self.backend == "nccl" else self.control


def init_group(dist, torch, backend, dev, allow_fallback, timeout_s=180.0):
    """Join the job.  Every rank first joins a gloo group (the rendezvous torch.distributed.run or this script's own
    launcher described in the environment); RCCL is then brought up as a second group on the same store and proved
    with one all-reduce of ones on the device, whose result is the number of ranks RCCL reached.  Whether that
    worked is agreed over gloo (MIN of the per-rank flags), so a failure on ONE rank is seen by all of them within
    `timeout_s` and nobody is left waiting in a collective.  If RCCL failed anywhere: exit code 3 on every rank,
    unless the caller allowed the fallback, in which case the timed barrier / MAX run over gloo."""
    import datetime
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(timeout_s, 60.0) + 120.0))
    control = dist.group.WORLD
    world = dist.get_world_size()
    if backend == "gloo":
        return Group(control, None, "gloo", 0)
    ok, err, device_group, reached = 1, "", None, 0
    try:
        kw = dict(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        try:
            device_group = dist.new_group(device_id=dev, **kw)
        except TypeError:  # a torch without new_group(device_id=)
            device_group = dist.new_group(**kw)
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(ones, group=device_group)
        torch.cuda.synchronize()
        reached = int(ones.item())
        if reached != world:
            raise RuntimeError("RCCL all-reduce of ones over %d ranks returned %d" % (world, reached))
    except Exception as e:  # noqa: BLE001 - any failure of the RCCL bring-up, on this rank
        ok, err = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        print("bench: RCCL bring-up failed on rank %d (%s)" % (dist.get_rank(), err), file=sys.stderr, flush=True)
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=control)
    if int(flag.item()) == 1:
        return Group(control, device_group, "nccl", reached)
    if not allow_fallback:
        print("bench: RCCL did not come up on every rank and --allow-gloo-fallback was not given: failing",
              file=sys.stderr, flush=True)
        dist.barrier(group=control)  # every rank has printed before the first one leaves
        os._exit(3)  # not sys.exit: a rank stuck half-way in an RCCL communicator must not block interpreter shutdown
    return Group(control, None, "gloo", 0, note="RCCL bring-up failed%s; --allow-gloo-fallback" % ((" (" + err + ")") if err else ""))


def synthetic_motion_tables(torch, dev, num_motions=2, frames=300, fps=30.0, seed=0):
    """Per-frame tables of `num_motions` synthetic clips in the layout igx_motion_tables takes (the reference's
    motion files are not in its tree): smooth random rotations and positions, 15 bodies, 28 joint velocities."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    F = frames * num_motions

    def smooth(*shape):
        t = torch.linspace(0, 6.28, frames).repeat(num_motions).view(F, *([1] * len(shape)))
        a, b = torch.randn(1, *shape, generator=g), torch.randn(1, *shape, generator=g)
        return torch.sin(t * (1 + a.abs())) * 0.5 + b * 0.3

    def unit(q):
        return q / q.norm(dim=-1, keepdim=True)

    ident = torch.tensor([0.0, 0.0, 0.0, 1.0])
    tabs = dict(
        motion_length=torch.full((num_motions,), (frames - 1) / fps, dtype=torch.float64),
        motion_dt=torch.full((num_motions,), 1.0 / fps, dtype=torch.float64),
        motion_num_frames=torch.full((num_motions,), frames, dtype=torch.int64),
        frame_offset=torch.arange(num_motions, dtype=torch.int64) * frames,
        root_pos=smooth(3) + torch.tensor([0.0, 0.0, 0.9]), root_rot=unit(smooth(4) + ident),
        local_rot=unit(smooth(15, 4) + ident), root_vel=smooth(3), root_ang_vel=smooth(3),
        key_pos=smooth(4, 3), dof_vel=smooth(28))
    return {k: v.contiguous().to(dev) for k, v in tabs.items()}


def extra_configs(torch, dist, ops, synthetic, dev, n, steps, group, rank, world):
    """The other configurations of the path, each timed with HIP events over `steps` launches on rotating input
    sets (per rank; the slowest rank is reported when there are several).  For world > 1 the Humanoid shard also
    runs the episode-statistics kernel every step and all-gathers the totals afterwards (configuration 3:
    8 x 2^20 Humanoid environments with an RCCL all-gather of episode statistics)."""
    from isaacgym_amd.episode_stats import EpisodeStats

    WARMUP = 120  # the first ~75 launches of a kernel on freshly allocated buffers run 2-9 % slower than the ones
                  # after them (tools/step_trend.py: Humanoid 182 / 172 / 168.5 us in launches 26-50 / 51-75 / 76-100,
                  # 167.5 from there on): the clock starts in the steady state

    def timed(fn, k=steps, warmup=WARMUP):
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(k):
            fn(warmup + i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / k * 1e3
        if group is not None and world > 1:
            t = torch.tensor([us], dtype=torch.float64, device=dev if group.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group.data)
            us = float(t.item())
        return us

    res = {}

    def unit_rows(q):
        return q / q.norm(dim=-1, keepdim=True)

    def report(name, us, units, bytes_per_unit, what):
        r = {"us": round(us, 2), "units": units, "units_per_s": units / us * 1e6, "what": what}
        if bytes_per_unit:
            gbps = units * bytes_per_unit / us / 1e3
            r.update(bytes_per_unit=bytes_per_unit, GBps=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBPS, 4))
        res[name] = r

    # Humanoid (configuration 3's per-GPU shard)
    p = synthetic.loco_params(synthetic.HUMANOID, seed=77 + rank, rng_offset=0)
    sets = [synthetic.loco_state(synthetic.HUMANOID, n, 500 * rank + s, dev) for s in range(NUM_INPUT_SETS)]
    for s in sets[1:]:
        for k in ("potentials", "prev_potentials", "reset", "progress", "randomize", "obs", "up_vec", "heading_vec", "rew"):
            s[k] = sets[0][k]
    caches = [dict() for _ in sets]

    def hum(i):
        p.rng_offset = i
        ops.humanoid_step(p, sets[i % NUM_INPUT_SETS], n, caches[i % NUM_INPUT_SETS])
    report("humanoid_step", timed(hum), n, 940, "Humanoid fused post_physics_step, %d envs per GPU" % n)
    if world > 1:
        es = EpisodeStats(n, dev)

        def hum_es(i):
            hum(i)
            es.step(sets[0]["rew"], sets[0]["reset"])
        us = timed(hum_es)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = es.totals(group.data)
        gather_us = (time.perf_counter() - t0) * 1e6
        report("humanoid_step+episode_stats", us, n, 940 + 28,
               "Humanoid step + episode-statistics kernel per step on every rank; totals all-gathered once over %s "
               "(%d ranks, 32 B per rank, %.0f us on rank 0 incl. the read-back; %d episodes finished)"
               % (group.backend, world, gather_us, tot[0]))
    del sets, caches

    # AllegroHand (configuration 4)
    pa = synthetic.allegro_params("full_state", 1, 0)
    sets = [synthetic.allegro_state(n, 300 * rank + s, dev) for s in range(3)]
    caches = [dict() for _ in sets]
    report("allegro_post", timed(lambda i: ops.allegro_post_physics(pa, sets[i % 3], n, caches[i % 3])), n, 748,
           "AllegroHand post_physics_step (obs + compute_hand_reward + global sums), %d envs" % n)
    caches = [dict() for _ in sets]
    report("allegro_pre", timed(lambda i: ops.allegro_pre_physics(pa, sets[i % 3], n, caches[i % 3])), n, 272,
           "AllegroHand pre_physics_step (flagged resets + action path), %d envs" % n)
    del sets, caches

    # HumanoidAMP step (configuration 5's environment side)
    pm = synthetic.amp_params(2)
    sets = [synthetic.amp_state(n, 200 * rank + s, dev, 2) for s in range(2)]
    caches = [dict() for _ in sets]
    report("humanoid_amp_step", timed(lambda i: ops.humanoid_amp_step(pm, sets[i % 2], n, caches[i % 2])), n, 2272,
           "HumanoidAMP post_physics_step (obs + reset + AMP history shift, 2 history steps), %d envs; 2960 B/env "
           "if the whole 780-B body rows are counted" % n)
    # reset_idx of the environments a masked list names (what the host-free reset hands over: entry i = environment i, -1
    # = not reset), 0.3 % of them set: new state in, observation and every history slot out
    gen = torch.Generator(device=dev).manual_seed(17 + rank)
    every = torch.arange(n, device=dev)
    masked = torch.where(torch.rand(n, device=dev, generator=gen) < 0.003, every, torch.full_like(every, -1))
    f32 = dict(dtype=torch.float32, device=dev)
    state = dict(root_pos=torch.randn(n, 3, generator=gen, **f32), root_rot=unit_rows(torch.randn(n, 4, generator=gen, **f32)),
                 dof_pos=torch.randn(n, 28, generator=gen, **f32), root_vel=torch.randn(n, 3, generator=gen, **f32),
                 root_ang_vel=torch.randn(n, 3, generator=gen, **f32), dof_vel=torch.randn(n, 28, generator=gen, **f32))
    flagged = int((masked >= 0).sum())
    report("humanoid_amp_reset_masked", timed(lambda i: ops.humanoid_amp_reset(pm, sets[i % 2], masked, state)), flagged, None,
           "HumanoidAMP reset_idx through a masked list over %d environments, %d of them reset per launch (state in, "
           "observation and AMP history out): resets/s" % (n, flagged))
    del sets, caches, state, masked, every

    # the torch_jit_utils functions the path names, one launch each on 2^22 rows into preallocated outputs (three
    # rotating input sets: 2^22 rows of three sets do not stay in the 256 MB last-level cache)
    from isaacgym_amd import _lib
    m = 1 << 22
    gen = torch.Generator(device=dev).manual_seed(9 + rank)

    def unit_quats():
        q = torch.randn(m, 4, device=dev, generator=gen)
        return q / q.norm(dim=-1, keepdim=True)

    Q = [(unit_quats(), unit_quats()) for _ in range(3)]
    V = [torch.randn(m, 3, device=dev, generator=gen) for _ in range(3)]
    T = [torch.rand(m, 1, device=dev, generator=gen) for _ in range(3)]
    X = [torch.randn(m, 8, device=dev, generator=gen) for _ in range(3)]
    lo8, hi8 = -torch.rand(8, device=dev, generator=gen) - 0.1, torch.rand(8, device=dev, generator=gen) + 0.1
    o4, o3, o8 = torch.empty(m, 4, device=dev), torch.empty(m, 3, device=dev), torch.empty(m, 8, device=dev)
    for name, nbytes, fn in (
            ("quat_mul", 48, lambda i: _lib.call_rowwise("igx_quat_mul", list(Q[i % 3]), o4, m)),
            ("quat_conjugate", 32, lambda i: _lib.call_rowwise("igx_quat_conjugate", [Q[i % 3][0]], o4, m)),
            ("quat_rotate", 40, lambda i: _lib.call_rowwise("igx_quat_rotate", [Q[i % 3][0], V[i % 3]], o3, m)),
            ("slerp", 52, lambda i: _lib.call_rowwise("igx_slerp", [Q[i % 3][0], Q[i % 3][1], T[i % 3]], o4, m)),
            ("scale", 64, lambda i: _lib.call_affine("igx_scale", X[i % 3], lo8, hi8, o8, m, 8)),
            ("unscale", 64, lambda i: _lib.call_affine("igx_unscale", X[i % 3], lo8, hi8, o8, m, 8)),
            ("tensor_clamp", 64, lambda i: _lib.call_affine("igx_tensor_clamp", X[i % 3], lo8, hi8, o8, m, 8))):
        report("torch_jit_utils." + name, timed(fn), m, nbytes, "torch_jit_utils.%s, %d rows per launch" % (name, m))
    del Q, V, T, X, o4, o3, o8

    # what VecTask.step does around a task's kernels when they do not fold it in: the action stage (clamp, self.actions,
    # forces: tasks/base/vec_task.py:381-384 + tasks/ant.py:368-374) and the tail (clamped copy + time-outs, :404-416)
    acts = [torch.randn(n, 8, device=dev, generator=gen) for _ in range(4)]
    kept, forces = torch.empty(n, 8, device=dev), torch.empty(n, 8, device=dev)
    gears = torch.rand(8, device=dev, generator=gen) * 100
    report("action_stage_ant", timed(lambda i: ops.action_stage(acts[i % 4], forces, "force", gain=gears, power_scale=1.0,
                                                                clip_actions=1.0, actions_out=kept)), n, 96,
           "VecTask.step head + Ant.pre_physics_step: clamped actions kept, joint forces out, %d envs x 8 actions" % n)
    obs4 = [torch.randn(n, 60, device=dev, generator=gen) for _ in range(4)]
    obs_out = torch.empty(n, 60, device=dev)
    prog = torch.randint(0, 1000, (n,), device=dev, generator=gen)
    rst = torch.randint(0, 2, (n,), device=dev, generator=gen)
    touts = torch.zeros(n, dtype=torch.bool, device=dev)
    report("vec_task_tail_ant", timed(lambda i: ops.vec_task_tail(obs4[i % 4], obs_out, 5.0, prog, rst, touts, 1000)), n,
           60 * 8 + 17, "VecTask.step tail on its own: clamped observation copy + time-out flags, %d envs x 60 columns" % n)
    del acts, kept, forces, obs4, obs_out, prog, rst, touts

    # Cartpole (configuration 1's task on the GPU) and the tasks either side of the path's five: ShadowHand,
    # HumanoidImitate (2^19 environments: their rows are twice as wide), AntImitate with its exact quantile
    pc = synthetic.cartpole_params(1, 0)
    sets = [synthetic.cartpole_state(n, 100 * rank + s, dev) for s in range(4)]
    caches = [dict() for _ in sets]
    report("cartpole_step", timed(lambda i: ops.cartpole_step(pc, sets[i % 4], n, caches[i % 4])), n, 68,
           "Cartpole observations + compute_cartpole_reward + resets, %d envs" % n)
    # the fork's sampling mode on top of it (tasks/cartpole_sample.py:99-121): two observation columns of every
    # environment per step, gathered on the device and streamed to pinned host memory on the recorder's side stream
    from isaacgym_amd import _lib
    from isaacgym_amd.recorder import TrajectoryRecorder as ColumnRecorder
    caches = [dict() for _ in sets]
    rsteps, rwarm = 32, 8
    crec = ColumnRecorder(n, 4, {"x": 0, "theta": 2}, rsteps + rwarm, dev)

    def step_and_capture(i):
        ops.cartpole_step(pc, sets[i % 4], n, caches[i % 4])
        crec.capture(sets[i % 4]["obs"])
    us = timed(step_and_capture, rsteps, rwarm)
    t0 = time.perf_counter()
    _lib.check(_lib.lib().igx_recorder_sync(crec._h))
    drain_ms = (time.perf_counter() - t0) * 1e3
    report("cartpole_step+sample_capture", us, n, None,
           "Cartpole step + capture of the x and theta columns of all %d environments (8 B per environment and step to "
           "pinned host memory): time per step on the producer stream; %d steps = %.1f MB on the host %.2f ms after the "
           "last step" % (n, crec.steps, 2 * 4 * n * crec.steps / 1e6, drain_ms))
    crec.close()
    del sets, caches
    half = max(n // 2, 1)
    ps = synthetic.shadow_params("full_state", 1, 0)
    sets = [synthetic.shadow_state(half, 400 * rank + s, dev) for s in range(3)]
    caches = [dict() for _ in sets]
    report("shadow_post", timed(lambda i: ops.shadow_post_physics(ps, sets[i % 3], half, caches[i % 3])), half, 1732,
           "ShadowHand post_physics_step (full_state observations + compute_hand_reward + global sums), %d envs" % half)
    del sets, caches
    ph = synthetic.him_params(1, 0)
    sets = [synthetic.him_state(half, 600 * rank + s, dev) for s in range(2)]
    report("humanoid_imitate_step", timed(lambda i: ops.humanoid_imitate_step(ph, sets[i % 2], half)), half, 1928,
           "HumanoidImitate post_physics_step (observations + imitation reward + resets), %d envs" % half)
    del sets
    from isaacgym_amd import _abi
    pi = synthetic.loco_params(synthetic.ANT, 1, 0)
    sets = [synthetic.loco_state(synthetic.ANT, n, 700 * rank + s, dev) for s in range(3)]
    lo, hi = (torch.tensor(v, device=dev, dtype=torch.float64) for v in synthetic.loco_limits(8))
    clip = lo + (hi - lo) * torch.rand(64, 8, device=dev, dtype=torch.float64)
    frame = torch.randint(0, 60, (n,), device=dev)
    pose_diff = torch.zeros(n, dtype=torch.float64, device=dev)
    qws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=dev)
    im = ops.imitate_args(clip, frame, pose_diff, qws, 0.99, 1.8, -2.0)
    report("ant_imitate_step", timed(lambda i: ops.ant_imitate_step(pi, sets[i % 3], im, n)), n, 556 + 32,
           "AntImitate post_physics_step: observations, exact 0.99 quantile of the pose differences over all %d envs "
           "(torch.quantile), reward and resets" % n)
    del sets

    # motion library: fetch_amp_obs_demo's kernel (configuration 5: 1 024 trajectories per fetch)
    tabs = synthetic_motion_tables(torch, dev)
    tables = ops.motion_tables(tabs)
    for m in (1024, 1 << 20):
        gen = torch.Generator(device=dev).manual_seed(m + rank)
        ids = torch.randint(0, 2, (m,), device=dev, generator=gen)
        times = (torch.rand(m, device=dev, dtype=torch.float64, generator=gen) * tabs["motion_length"][ids]).contiguous()
        out = torch.empty(m, 105, device=dev)
        report("motion_amp_obs_%d" % m, timed(lambda i: ops.motion_amp_obs(tables, ids, times, False, out)), m, None,
               "motion_lib frame blend + slerp + local-rotation-to-dof + build_amp_observations, %d samples per launch "
               "(table gathers are cache resident: samples/s, no roofline)" % m)

    # a whole fetch_amp_obs_demo without the host (configuration 5: 512 trajectories x 2 history steps = 1 024 samples):
    # the two draws on the device, then one launch that tiles over the history steps; and the same with every batch
    # streamed to pinned host memory by the whole-row recorder on its side stream
    from isaacgym_amd.recorder import TrajectoryRecorder
    nb, K = 512, 2
    cdf = torch.tensor([0.5, 1.0], dtype=torch.float64, device=dev)
    d_ids = torch.empty(nb, dtype=torch.int64, device=dev)
    d_times = torch.empty(nb, dtype=torch.float64, device=dev)
    batch = torch.empty(nb, K * 105, device=dev)

    def fetch(i):
        ops.motion_fetch_amp_obs(tables, cdf, nb, K, 0.0332, False, batch, d_ids, d_times, seed=5 + rank, rng_offset=i)
    report("amp_demo_fetch_device_sampling", timed(fetch, 4 * steps), nb * K, None,
           "fetch_amp_obs_demo with device-side draws in one launch (igx_motion_fetch_amp_obs: the draws of "
           "igx_motion_sample and the batch of igx_motion_amp_obs_steps), %d trajectories x %d history steps per fetch; "
           "no host transfer, no synchronisation" % (nb, K))
    if world > 1:
        # the imitation batches of all ranks as one demo buffer: every fetch followed by an all-gather of the
        # [512, 210] batch over the data group (RCCL when it is up)
        from isaacgym_amd import distributed

        def fetch_and_gather(i):
            fetch(i)
            return distributed.all_gather_rows(batch, group.data)
        try:
            us = timed(fetch_and_gather, 2 * steps)
            rows = fetch_and_gather(0).shape[0]
            report("amp_demo_fetch+all_gather", us, nb * K * world, None,
                   "the same fetch on every rank followed by an all-gather of the [%d, %d] batches over %s: %d rows per "
                   "gathered demo buffer, %d ranks (time per fetch + gather on the slowest rank, host clock of the "
                   "enqueueing thread excluded: HIP events)" % (nb, K * 105, group.backend, rows, world))
        except Exception as e:  # noqa: BLE001 - an extra line of the report must not cost the run its result
            res["amp_demo_fetch+all_gather"] = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
    rec = TrajectoryRecorder(nb, K * 105, None, 4 * steps + WARMUP, dev)

    def fetch_and_stream(i):
        fetch(i)
        rec.capture(batch)
    us = timed(fetch_and_stream, 4 * steps)
    t0 = time.perf_counter()
    host_batches = rec.finish()
    drain_ms = (time.perf_counter() - t0) * 1e3
    report("amp_demo_fetch_streamed_to_host", us, nb * K, None,
           "the same with every [%d, %d] batch captured by the whole-row recorder (pinned host memory, side stream): "
           "time per fetch on the producer stream; %d batches = %.1f MB on the host %.2f ms after the last fetch"
           % (nb, K * 105, host_batches.shape[0], host_batches.numel() * 4 / 1e6, drain_ms))
    rec.close()
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # nothing above has touched the GPU
    # standard output carries ONE JSON line and nothing else: libraries that write to file descriptor 1 themselves
    # (gloo announces its connections there) are sent to standard error for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from isaacgym_amd import ops, synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev_index = local_rank if args.all_ranks_on_device < 0 else args.all_ranks_on_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    grouped = world > 1 or args.force_group
    if grouped:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        group = init_group(dist, torch, args.backend, dev, args.allow_gloo_fallback)
    backend = group.backend if grouped else None

    spec = synthetic.ANT if args.task == "ant" else synthetic.HUMANOID
    step = ops.ant_step if args.task == "ant" else ops.humanoid_step
    n = args.envs_per_gpu
    p = synthetic.loco_params(spec, seed=1234 + rank, rng_offset=0)

    # independent input sets (each rank: its own shard, seeded by rank)
    sets = [synthetic.loco_state(spec, n, 1000 * rank + s, dev) for s in range(NUM_INPUT_SETS)]
    # persistent per-env state is shared across the rotating "physics" inputs
    for s in sets[1:]:
        for k in ("potentials", "prev_potentials", "reset", "progress", "randomize", "obs", "up_vec",
                  "heading_vec", "rew"):
            s[k] = sets[0][k]

    # the tasks' own calling convention: one validated argument struct per buffer set, reused from step to step
    caches = [dict() for _ in sets]

    es = None
    if args.episode_stats:
        from isaacgym_amd.episode_stats import EpisodeStats
        es = EpisodeStats(n, dev)

    def run(k0, k, with_stats=True):
        for i in range(k0, k0 + k):
            p.rng_offset = i
            step(p, sets[i % NUM_INPUT_SETS], n, caches[i % NUM_INPUT_SETS])
            if es is not None and with_stats:
                es.step(sets[0]["rew"], sets[0]["reset"])

    def barrier():
        if grouped:
            dist.barrier(group=group.data)  # RCCL: an all-reduce on the device; gloo: host
        torch.cuda.synchronize()

    run(0, args.settle)  # steady state first: what a training job spends all but its first milliseconds in
    run(args.settle, args.warmup)
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.settle + args.warmup, args.steps)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0  # this rank's K steps; the MAX over the ranks is taken below
    barrier()
    if es is not None:
        # the episode kernels sit between ev0 and ev1: the step kernel is priced by K more launches of its own, after
        # the timed region (an event pair around every step of the timed loop would cost that loop a queue drain per
        # event: measured 0.94 instead of 0.20 ms per step)
        k0 = args.settle + args.warmup + args.steps
        ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev2.record()
        run(k0, args.steps, with_stats=False)
        ev3.record()
        torch.cuda.synchronize()
        kernel_ms = ev2.elapsed_time(ev3) / args.steps
    else:
        kernel_ms = ev0.elapsed_time(ev1) / args.steps  # device time per launch on the launch stream

    if grouped:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group.data)
        elapsed = float(tt.item())

    # one 32-byte-per-rank all-gather (over RCCL when it is up and world > 1)
    stats = es.totals(group.data if grouped else None) if es is not None else None

    configs = None
    if not args.no_configs and args.task == "ant":
        del sets, caches
        torch.cuda.empty_cache()
        configs = extra_configs(torch, dist, ops, synthetic, dev, n, args.config_steps, group, rank, world)

    if rank == 0:
        total_env_steps = float(n) * world * args.steps
        value = total_env_steps / elapsed
        bytes_per_env = ANT_BYTES_PER_ENV_STEP if args.task == "ant" else 940
        achieved = n * bytes_per_env / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC if args.task == "ant" else METRIC.replace("Ant", args.task.capitalize()),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s obs/reward/reset fused post_physics_step, %d envs per GPU, AoS reference "
                                   "layout, in-kernel Philox resets" % (args.task, n),
                       "envs_per_gpu": n, "global_envs": n * world, "input_sets": NUM_INPUT_SETS, "settle_launches": args.settle,
                       "parallelism": "env-shard x%d (no data-path collective)" % world,
                       "collective_backend": backend},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": load_traffic() if args.task == "ant" and n == ENVS_PER_GPU else None,
                         "traffic_source": ("%s (committed rocprofv3 --pmc run of this command, not measured by this run)"
                                            % TRAFFIC_FILE) if args.task == "ant" and n == ENVS_PER_GPU else None,
                         "kernel_ms": kernel_ms, "bytes_per_env_step": bytes_per_env},
        }
        if grouped:
            out["rccl_ranks"] = group.rccl_ranks
            out["collectives"] = {"rendezvous_and_agreement": "gloo", "barrier": backend, "max_of_elapsed": backend}
            if es is not None or (configs and "humanoid_step+episode_stats" in configs):
                out["collectives"]["episode_stats_all_gather"] = backend
            if configs and "us" in configs.get("amp_demo_fetch+all_gather", {}):
                out["collectives"]["imitation_batch_all_gather"] = backend
            if group.note:
                out["collectives"]["note"] = group.note
        if configs is not None:
            out["configs"] = configs
        if stats is not None:
            out["episode_stats"] = {"episodes_finished": stats[0], "mean_reward": stats[1] / max(stats[0], 1),
                                    "mean_length": stats[2] / max(stats[0], 1), "steps": stats[3],
                                    "note": "extra kernel inside the timed region; totals all-gathered over the ranks"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, n, args.task)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    if grouped:
        # the result is out; a teardown of the communicators that does not return must not turn it into a failure
        import threading

        def give_up():
            print("bench: process-group teardown still running after 60 s: exiting", file=sys.stderr, flush=True)
            os._exit(0)
        watchdog = threading.Timer(60.0, give_up)
        watchdog.daemon = True
        watchdog.start()
        dist.barrier(group=group.control)
        dist.destroy_process_group()
        watchdog.cancel()


if __name__ == "__main__":
    main()
