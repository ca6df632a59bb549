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

Rank 0 prints ONE JSON line.  `value` is whole-job env-steps/s with all inputs
resident in HBM.  `roofline` prices the dominant (only) kernel against the HBM
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
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--task", default="ant", choices=["ant", "humanoid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
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
    out = {"unit": "env-steps/s", "kind": "port", "host_cpus": host_cpus}

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
        head = out["torch_ref"]
    else:
        head = out["port"]
    out.update(value=head["value"], cores=head["cores"], sample=head["sample"])
    return out


def load_traffic():
    """HBM bytes per launch from the committed PMC summary, if there is one."""
    path = os.path.join(ROOT, "profiles", "traffic_ant.json")
    try:
        with open(path) as f:
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


def init_group(dist, torch, backend, dev):
    """Join the job's process group.  The collectives of this benchmark are a barrier and one MAX of a scalar, so
    when RCCL cannot come up on this node (it raises on every rank alike) the ranks meet again over gloo on the
    next port instead of losing the measurement; the JSON line names the backend that was used."""
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
            return "nccl"
        except Exception as e:  # noqa: BLE001 - any failure of the RCCL bring-up
            print("bench: RCCL bring-up failed (%s: %s); using gloo for the barrier" % (type(e).__name__, e),
                  file=sys.stderr, flush=True)
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    dist.init_process_group("gloo")
    return "gloo"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # nothing above has touched the GPU
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
    backend = args.backend
    grouped = world > 1 or args.force_group
    if grouped:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        backend = init_group(dist, torch, args.backend, dev)

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

    step_events = []  # with --episode-stats: one event pair per step around the step kernel alone

    def run(k0, k, timed=False):
        for i in range(k0, k0 + k):
            p.rng_offset = i
            if es is not None and timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                step(p, sets[i % NUM_INPUT_SETS], n, caches[i % NUM_INPUT_SETS])
                e1.record()
                step_events.append((e0, e1))
            else:
                step(p, sets[i % NUM_INPUT_SETS], n, caches[i % NUM_INPUT_SETS])
            if es is not None:
                es.step(sets[0]["rew"], sets[0]["reset"])

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, args.warmup)
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.warmup, args.steps, timed=True)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0  # this rank's K steps; the MAX over the ranks is taken below
    barrier()
    if step_events:  # the episode kernels sit between ev0 and ev1: price the step kernel by its own pairs
        kernel_ms = sum(a.elapsed_time(b) for a, b in step_events) / len(step_events)
    else:
        kernel_ms = ev0.elapsed_time(ev1) / args.steps  # device time per launch on the launch stream

    if grouped:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stats = es.totals() if es is not None else None  # one 32-byte-per-rank all-gather (RCCL when world > 1)

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
                       "envs_per_gpu": n, "global_envs": n * world, "input_sets": NUM_INPUT_SETS,
                       "parallelism": "env-shard x%d (no data-path collective)" % world,
                       "collective_backend": backend if grouped else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": load_traffic() if args.task == "ant" and n == ENVS_PER_GPU else None,
                         "kernel_ms": kernel_ms, "bytes_per_env_step": bytes_per_env},
        }
        if stats is not None:
            out["episode_stats"] = {"episodes_finished": stats[0], "mean_reward": stats[1] / max(stats[0], 1),
                                    "mean_length": stats[2] / max(stats[0], 1), "steps": stats[3],
                                    "note": "extra kernel inside the timed region; totals all-gathered over the ranks"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, n, args.task)
        print(json.dumps(out), flush=True)

    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
