"""Child of tests/test_gpu_multiprocess.py: one rank of a two-rank job whose ranks share GPU 0 and talk over
gloo.  The PRODUCT entry points (libigx kernels) run the shards; rank 0 also runs the unsharded call and
compares."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd import _abi, distributed, ops, synthetic  # noqa: E402

DEV = "cuda:0"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")


def shard_of(full, start, count, rows_per_env=3):
    out = {}
    for k, v in full.items():
        if v is None:
            out[k] = None
        elif k == "root_state_tensor":
            out[k] = v[rows_per_env * start:rows_per_env * (start + count)].clone()
        elif k in ("object_indices", "goal_object_indices"):
            out[k] = (v[start:start + count] - rows_per_env * start).contiguous()
        elif k in ("workspace", "consecutive_successes"):
            out[k] = v.clone()
        else:
            out[k] = v[start:start + count].clone()
    return out


def hands(which):
    n, steps, interval = 20011, 5, 2
    if which == "allegro":
        p = synthetic.allegro_params("full_state", seed=1, rng_offset=2)
        full = synthetic.allegro_state(n, 8, DEV, "full_state")
        nrand, post, shard, fin = 45, ops.allegro_post_physics, ops.allegro_post_physics_shard, ops.allegro_finalize_shards
    else:
        p = synthetic.shadow_params("full_state", seed=1, rng_offset=2)
        full = synthetic.shadow_state(n, 8, DEV, "full_state")
        nrand, post, shard, fin = 61, ops.shadow_post_physics, ops.shadow_post_physics_shard, ops.shadow_finalize_shards
    full["rand"] = torch.rand(n, nrand, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    full["consecutive_successes"].fill_(0.25)
    s, c = distributed.env_shard(n, rank, world)
    mine = shard_of(full, s, c)
    q = distributed.ShardRecordQueue(mine["consecutive_successes"], interval)
    apply = lambda records, k: fin(p, dict(consecutive_successes=q.pending), records, k)
    for step in range(steps):
        full["reset"][step::7] = 1
        mine["reset"].copy_(full["reset"][s:s + c])  # the same flags, in the shard's numbering
        mine["shard_record"] = q.slot()
        shard(p, mine, c)
        if q.push():
            q.flush(apply)
        post(p, full, n)
    q.drain(apply)
    torch.cuda.synchronize()
    for k in ("obs", "rew", "reset", "reset_goal", "progress", "successes"):
        assert torch.equal(mine[k], full[k][s:s + c]), (which, k)
    got, want = float(mine["consecutive_successes"]), float(full["consecutive_successes"])
    assert abs(got - want) <= 1e-6 * abs(want) and want != 0.25, (got, want)
    both = [None] * world
    dist.all_gather_object(both, mine["consecutive_successes"].cpu().numpy().tobytes())
    assert both[0] == both[1]  # the same bits on every rank
    return got


def quantile():
    n = 300001
    x = torch.randn(n, dtype=torch.float64, generator=torch.Generator().manual_seed(5)) ** 3
    s, c = distributed.env_shard(n, rank, world)
    ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)

    def all_reduce(hist):  # gloo: through the host
        h = hist.cpu()
        dist.all_reduce(h)
        hist.copy_(h)
    got = float(ops.quantile_sharded(x[s:s + c].to(DEV).contiguous(), 0.99, n, ws, all_reduce))
    assert got == float(torch.quantile(x, 0.99)), got
    return got


def window_quantile():
    """The two-collective protocol on the product's kernels; every step against torch.quantile and the six-pass select."""
    n = 400003
    s, c = distributed.env_shard(n, rank, world)
    wq = ops.WindowQuantile(DEV)
    ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    gen = torch.Generator().manual_seed(6)
    counts = []
    for step in range(6):
        x = torch.randn(n, dtype=torch.float64, generator=gen).abs() * (1.0 + 0.01 * step)
        if step == 4:
            x[::2] = 1.25  # far more copies of one value than the exchange carries
        mine = x[s:s + c].to(DEV).contiguous()
        q = 0.5 if step >= 3 else 0.99
        ok = wq.run(mine, q, n, distributed.all_reduce_sum, distributed.gather_records, out=ws.view(torch.float64)[5:6])
        want = float(torch.quantile(x, q))
        assert ok and float(wq.result()) == want == float(ws.view(torch.float64)[5]), (step, float(wq.result()), want)
        assert float(ops.quantile_sharded(mine, q, n, ws, distributed.all_reduce_sum)) == want
        counts.append(wq.collectives)
    assert counts[1] == counts[2] == 2 and counts[4] > 2, counts
    return counts


def ant_imitate_class(protocol):
    """The AntImitate class on the reference's recording, each rank with its half of the environments: the
    global threshold, flags and target frames are those of the unsharded recording."""
    from isaacgym_amd import cfg as C, tasks
    from isaacgym_amd.sim import ReplaySim
    from tests import helpers as H
    g = H.load_golden("ant_imitate")
    N, T = g["meta"]["N"], g["meta"]["T"]
    s, c = distributed.env_shard(N, rank, world)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    cut = lambda a: a[s:s + c]
    dofs = lambda a: np.ascontiguousarray(a.reshape(N, 8, 2)[s:s + c])
    frames = [dict(root_states=cut(g["in_root"][t]), dof_state=dofs(g["in_dof_state"][t]).reshape(c * 8, 2),
                   sensors=cut(g["in_sensors"][t])) for t in range(T)]
    init = dict(root_states=cut(g["in_root"][0]), dof_state=dofs(g["in_dof_state"][0]), sensors=cut(g["in_sensors"][0]))
    sim = ReplaySim(c, DEV, init, frames, dict(dof_limits_lower=g["dof_limits_lower"].tolist(),
                                               dof_limits_upper=g["dof_limits_upper"].tolist()))
    task = tasks.AntImitate(C.task_config("AntImitate", c, quantileCollectives=protocol), DEV, DEV, sim=sim, motion=g["motion"])
    task.initial_root_states.copy_(dev(cut(g["initial_root_states"])))
    task.initial_dof_pos.copy_(dev(cut(g["initial_dof_pos"])))
    task.progress_buf.copy_(dev(cut(g["progress0"])))
    task.potentials.copy_(dev(cut(g["potentials0"])))
    for t in range(T):
        task.reset_buf.copy_(dev(cut(g["in_reset"][t])))
        task.reset_rand = dev(cut(g["rand"][t]))
        _, rew, reset, _ = task.step(dev(cut(g["in_actions"][t])))
        assert task.pose_threshold == float(g["threshold"][t]), (protocol, t, task.pose_threshold, float(g["threshold"][t]))
        assert np.array_equal(reset.cpu().numpy(), cut(g["reset"][t]))
        assert np.array_equal(task.progress_buf.cpu().numpy(), cut(g["progress"][t]))
        assert np.array_equal(task.target_pose_idx.cpu().numpy(), cut(g["target_pose_idx"][t]))
    if protocol == "window":
        assert task._window is not None and task._window.fallbacks == 0
    return T


out = {"ant_imitate": [ant_imitate_class("window"), ant_imitate_class("radix")], "allegro": hands("allegro"), "shadow": hands("shadow"), "quantile": quantile(), "window_quantile": window_quantile()}
dist.barrier()
if rank == 0:
    print(json.dumps(out))
dist.destroy_process_group()
