"""One process per GPU without an external launcher.

The reference's multi-GPU contract is only "rank -> device, seed offset by rank"
(utils/rlgames_utils.py:91-112, utils/utils.py:87-94); the processes themselves are
started by torchrun.  `spawn_ranks` is the same thing for a script that was started
as a single plain process: it starts `world` fresh children with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's standard output, and
fails if any child fails.

The parent never touches the GPU (no HIP call, no torch.cuda call): children are new
processes started with subprocess, never an exec from a process that has initialised
the device.
"""
import os
import socket
import subprocess
import sys
import time


def free_port(host="127.0.0.1"):
    """A TCP port that was free a moment ago on `host`."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None, addr="127.0.0.1"):
    """The environment of rank `rank` of a one-node job (what torchrun would set)."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": addr, "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return env


def spawn_ranks(argv, world, timeout=None, port=None, poll=0.05, env=None):
    """Run `argv` (a full command line) once per rank and wait.

    Returns (returncode, rank0_stdout).  returncode is 0 only if every rank exited 0;
    otherwise it is the code of the first rank seen failing (negative signals become
    128 + signal), and the ranks still running are terminated, then killed.  Standard
    output of the other ranks goes to this process's standard error, so that rank 0's
    output can be relayed unchanged.  124 when `timeout` seconds pass.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port() if port is None else port
    procs = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(
                list(argv), env=rank_env(r, world, port, env),
                stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=True))
    except Exception:
        _stop(procs)
        raise
    deadline = None if timeout is None else time.monotonic() + timeout
    failed = None
    out0 = []
    # rank 0's pipe is drained by communicate() at the end; its output here is one JSON line,
    # far below the pipe capacity, so polling without reading cannot block the child
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if deadline is not None and time.monotonic() > deadline:
            failed = (-1, 124)
            break
        time.sleep(poll)
    if failed is not None:
        _stop(procs)
    try:
        text, _ = procs[0].communicate(timeout=10)
        out0.append(text or "")
    except Exception:
        pass
    if failed is None:
        return 0, "".join(out0)
    rank, code = failed
    if rank >= 0:
        sys.stderr.write("rank %d exited with code %d; stopped the other ranks\n" % (rank, code))
    else:
        sys.stderr.write("ranks did not finish within %.0f s; stopped them\n" % timeout)
    return (128 - code if code < 0 else code), "".join(out0)


def _stop(procs, grace=5.0):
    """Terminate exactly the children started here (by pid), then kill what is left."""
    for p in procs:
        if p.poll() is None:
            try:
                p.terminate()
            except OSError:
                pass
    t_end = time.monotonic() + grace
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.02)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()
