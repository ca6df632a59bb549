"""The N > 1 layout on CPU: two gloo ranks, environments sharded, the product's
collective helpers, the CPU checker standing in for the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from isaacgym_amd import distributed, synthetic
from oracle import oracle


def test_env_shard_partitions_exactly():
    for n in (0, 1, 7, 64, 1 << 20, (1 << 20) + 5):
        for w in (1, 2, 3, 8):
            blocks = [distributed.env_shard(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        distributed.env_shard(10, 2, 2)
    assert len({distributed.shard_seed(5, r) for r in range(8)}) == 8 and distributed.shard_seed(5, 0) == 5


def test_single_process_is_a_world_of_one():
    assert distributed.world_size() == 1
    r = torch.arange(16, dtype=torch.uint8)
    assert torch.equal(distributed.gather_records(r), r.unsqueeze(0))
    assert distributed.all_gather_rows(torch.ones(3, 2)).shape == (3, 2)
    assert distributed.max_over_ranks(2.5, "cpu") == 2.5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _slice_state(t, start, count):
    """Rank's block of a global AllegroHand state (3 actor rows per environment)."""
    out = {}
    for k, v in t.items():
        if v is None:
            out[k] = None
        elif k == "root_state_tensor":
            out[k] = v[3 * start:3 * (start + count)].copy()
        elif k in ("object_indices", "goal_object_indices"):
            out[k] = v[start:start + count] - 3 * start
        elif k in ("consecutive_successes", "workspace"):
            out[k] = v.copy()
        else:
            out[k] = v[start:start + count].copy()
    return out


def _allegro_rank(rank, world, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    got = distributed.init_from_env(backend="gloo")
    assert got == (rank, rank, world) and distributed.world_size() == world
    oracle.set_num_threads(2)
    full = {k: (v.numpy().copy() if v is not None else None)
            for k, v in synthetic.allegro_state(n, 31, "cpu", "full_state", True).items()}
    full["rand"] = np.random.default_rng(1).random((n, 45), dtype=np.float32)
    start, count = distributed.env_shard(n, rank, world)
    mine = _slice_state(full, start, count)
    p = synthetic.allegro_params("full_state")
    for step in range(3):
        oracle.allegro_pre_physics(p, mine, count)
        oracle.allegro_post_physics_shard(p, mine, count)
        rec = torch.from_numpy(mine["workspace"].view(np.uint8)[:16].copy())
        records = distributed.gather_records(rec)
        assert records.shape == (world, 16)
        oracle.allegro_finalize_shards(p, mine, records.numpy())
    # episode statistics: one 32-byte record per rank, gathered and added in rank order
    from isaacgym_amd.episode_stats import sum_records
    rng = np.random.default_rng(5)
    rew, done = rng.standard_normal(n).astype(np.float32), (rng.random(n) < 0.2).astype(np.int64)
    k, sr, sl = oracle.episode_stats_step(dict(rew=rew[start:start + count].copy(), done=done[start:start + count].copy(),
                                               current_rewards=np.zeros(count, np.float32),
                                               current_lengths=np.zeros(count, np.float32),
                                               workspace=np.zeros(8, np.int64)), count)
    rec = torch.tensor([k, 0, 0, 1], dtype=torch.int64)
    rec[1:3] = torch.tensor([sr, sl], dtype=torch.float64).view(torch.int64)
    tot = sum_records(distributed.gather_records(rec))
    kf, srf, slf = oracle.episode_stats_step(dict(rew=rew.copy(), done=done.copy(), current_rewards=np.zeros(n, np.float32),
                                                  current_lengths=np.zeros(n, np.float32),
                                                  workspace=np.zeros(8, np.int64)), n)
    assert tot[0] == kf and tot[2] == slf and abs(tot[1] - srf) < 1e-9 * max(1.0, abs(srf)) and tot[3] == 1
    # imitation batches: per-rank rows concatenated in rank order
    rows = distributed.all_gather_rows(torch.full((4, 3), float(rank)))
    assert rows.shape == (4 * world, 3) and torch.equal(rows[::4, 0], torch.arange(world, dtype=torch.float32))
    assert distributed.max_over_ranks(float(rank), "cpu") == world - 1
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), start=start, count=count,
             **{k: v for k, v in mine.items() if v is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_allegro_sharded_over_two_gloo_ranks_equals_one_shard(tmp_path):
    n, world = 4001, 2
    mp.spawn(_allegro_rank, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    full = {k: (v.numpy().copy() if v is not None else None)
            for k, v in synthetic.allegro_state(n, 31, "cpu", "full_state", True).items()}
    full["rand"] = np.random.default_rng(1).random((n, 45), dtype=np.float32)
    p = synthetic.allegro_params("full_state")
    for step in range(3):
        oracle.allegro_pre_physics(p, full, n)
        oracle.allegro_post_physics(p, full, n)
    assert full["reset"].sum() > 0
    cons = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        s, c = int(z["start"]), int(z["count"])
        for k in ("obs", "states", "rew", "reset", "reset_goal", "progress", "successes", "goal_states", "dof_state",
                  "cur_targets"):
            assert np.array_equal(z[k].view(np.uint8), full[k][s:s + c].view(np.uint8)), (r, k)
        assert np.array_equal(z["root_state_tensor"], full["root_state_tensor"][3 * s:3 * (s + c)])
        cons.append(z["consecutive_successes"].copy())
    assert np.array_equal(cons[0].view(np.uint32), cons[1].view(np.uint32))  # same bits on every rank
    assert cons[0][0] != 0.0
    np.testing.assert_allclose(cons[0], full["consecutive_successes"], rtol=1e-6)


def _shadow_state(n):
    full = {k: (v.numpy().copy() if v is not None else None)
            for k, v in synthetic.shadow_state(n, 31, "cpu", "full", True).items()}
    full["rand"] = np.random.default_rng(2).random((n, 61), dtype=np.float32)
    return full


def _shadow_rank(rank, world, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    oracle.set_num_threads(2)
    start, count = distributed.env_shard(n, rank, world)
    mine = _slice_state(_shadow_state(n), start, count)
    p = synthetic.shadow_params("full")
    for step in range(3):
        oracle.shadow_pre_physics(p, mine, count)
        oracle.shadow_post_physics_shard(p, mine, count)
        records = distributed.gather_records(torch.from_numpy(mine["workspace"].view(np.uint8)[:16].copy()))
        oracle.shadow_finalize_shards(p, mine, records.numpy())
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), start=start, count=count,
             **{k: v for k, v in mine.items() if v is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_shadow_sharded_over_three_gloo_ranks_equals_one_shard(tmp_path):
    """ShadowHand (157-column policy observations + 211-column asymmetric states) over three uneven shards."""
    n, world = 1000, 3
    mp.spawn(_shadow_rank, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    full = _shadow_state(n)
    p = synthetic.shadow_params("full")
    for step in range(3):
        oracle.shadow_pre_physics(p, full, n)
        oracle.shadow_post_physics(p, full, n)
    assert full["reset"].sum() > 0
    cons = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        s, c = int(z["start"]), int(z["count"])
        for k in ("obs", "states", "rew", "reset", "reset_goal", "progress", "successes", "goal_states", "dof_state",
                  "cur_targets", "prev_targets"):
            assert np.array_equal(z[k].view(np.uint8), full[k][s:s + c].view(np.uint8)), (r, k)
        cons.append(z["consecutive_successes"].copy())
    assert all(np.array_equal(c.view(np.uint32), cons[0].view(np.uint32)) for c in cons)  # same bits on every rank
    np.testing.assert_allclose(cons[0], full["consecutive_successes"], rtol=1e-6)


def _quantile_rank(rank, world, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    x = np.random.default_rng(11).standard_normal(n) ** 3
    x[: n // 5] = 0.75  # ties across the shard boundary
    start, count = distributed.env_shard(n, rank, world)
    results = []
    for q in (0.99, 0.5, 0.0, 1.0):
        ws = oracle.quantile_workspace()
        oracle.quantile_begin(ws, q, n)          # every rank with the GLOBAL count
        for p in range(6):
            oracle.quantile_histogram(x[start:start + count], ws, p)
            hist = torch.from_numpy(oracle.quantile_hist_view(ws, p))   # a view: reduced in place
            dist.all_reduce(hist)
        results.append(oracle.quantile_finish(ws))
    np.save(os.path.join(outdir, "q%d.npy" % rank), np.array(results))
    dist.barrier()
    dist.destroy_process_group()


def test_quantile_sharded_over_two_gloo_ranks_equals_torch(tmp_path):
    """The one real exchange of the imitation path: per-pass histograms all-reduced over the ranks make
    every rank select the same order statistics as torch.quantile over all environments."""
    n, world = 50001, 2
    mp.spawn(_quantile_rank, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    x = np.random.default_rng(11).standard_normal(n) ** 3
    x[: n // 5] = 0.75
    want = np.array([float(torch.quantile(torch.from_numpy(x), q)) for q in (0.99, 0.5, 0.0, 1.0)])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "q%d.npy" % r))
        assert np.array_equal(got, want), (r, got, want)


def _window_rank(rank, world, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    from tests.helpers import oracle_window_quantile
    wq = oracle_window_quantile()
    start, count = distributed.env_shard(n, rank, world)
    rng = np.random.default_rng(12)
    rows = []
    for step in range(12):
        x = np.abs(rng.standard_normal(n)) * (1.0 + 0.02 * step)
        if step == 7:
            x[: n // 2] = 0.75  # more copies of one value than the exchange can carry
        if step == 9:
            x *= 1e5            # the value leaves the window
        ok = wq.run(torch.from_numpy(x[start:start + count].copy()), 0.5, n,
                    lambda t: distributed.all_reduce_sum(t), lambda t: distributed.gather_records(t))
        rows.append([float(wq.result()), float(ok), float(wq.collectives)])
    np.save(os.path.join(outdir, "w%d.npy" % rank), np.array(rows))
    dist.barrier()
    dist.destroy_process_group()


def test_window_quantile_over_two_gloo_ranks_equals_torch_in_two_collectives(tmp_path):
    """The two-collective form of the same exchange: the product's protocol (ops.WindowQuantile.run) with real
    gloo collectives over two processes; values bit-equal torch.quantile over all environments, and once the
    window sits on the value a call costs one all-reduce and one all-gather."""
    n, world = 60001, 2
    mp.spawn(_window_rank, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(12)
    want = []
    for step in range(12):
        x = np.abs(rng.standard_normal(n)) * (1.0 + 0.02 * step)
        if step == 7:
            x[: n // 2] = 0.75
        if step == 9:
            x *= 1e5
        want.append(float(torch.quantile(torch.from_numpy(x), 0.5)))
    got = [np.load(os.path.join(str(tmp_path), "w%d.npy" % r)) for r in range(world)]
    assert np.array_equal(got[0], got[1])
    assert np.array_equal(got[0][:, 0], np.array(want))
    assert np.all(got[0][:, 1] == 1.0)
    assert list(got[0][[2, 3, 4, 5, 6, 11], 2]) == [2.0] * 6, got[0][:, 2]
    assert got[0][7, 2] > 2, got[0][:, 2]  # shrinking the window down to the one repeated value


def _queued_rank(rank, world, port, n, outdir, which):
    """The same sharded run twice: the record exchange on every step, and queued (distributed.ShardRecordQueue:
    one gather per `interval` steps, applied in step order) - consecutive_successes must come out bit-identical."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    oracle.set_num_threads(2)
    start, count = distributed.env_shard(n, rank, world)
    if which == "allegro":
        full = {k: (v.numpy().copy() if v is not None else None)
                for k, v in synthetic.allegro_state(n, 31, "cpu", "full_state", True).items()}
        full["rand"] = np.random.default_rng(1).random((n, 45), dtype=np.float32)
        p = synthetic.allegro_params("full_state")
        pre, shard, fin = oracle.allegro_pre_physics, oracle.allegro_post_physics_shard, oracle.allegro_finalize_shards
    else:
        full = _shadow_state(n)
        p = synthetic.shadow_params("full")
        pre, shard, fin = oracle.shadow_pre_physics, oracle.shadow_post_physics_shard, oracle.shadow_finalize_shards
    steps, interval = 7, 3   # two full horizons and a partial one
    a = _slice_state(full, start, count)
    per_step = []
    for step in range(steps):
        a["reset"][step::5] = 1  # keep episodes finishing so the average keeps moving
        pre(p, a, count)
        shard(p, a, count)
        fin(p, a, distributed.gather_records(torch.from_numpy(a["workspace"].view(np.uint8)[:16].copy())).numpy())
        per_step.append(a["consecutive_successes"].copy())
    b = _slice_state(full, start, count)
    visible = torch.from_numpy(b["consecutive_successes"])
    q = distributed.ShardRecordQueue(visible, interval)
    apply = lambda records, k: fin(p, dict(consecutive_successes=q.pending.numpy()), records.numpy().view(np.uint8), k)
    seen = []
    for step in range(steps):
        b["reset"][step::5] = 1
        pre(p, b, count)
        b["shard_record"] = q.slot().numpy()
        shard(p, b, count)
        if q.push():
            q.flush(apply)
        seen.append(visible.numpy().copy())
    assert q.flushes == steps // interval
    # visible lags: after the flush at the end of horizon h it shows the state after horizon h - 1
    assert np.array_equal(seen[2], np.zeros(1, np.float32))
    assert np.array_equal(seen[5].view(np.uint32), per_step[2].view(np.uint32))
    q.drain(apply)
    assert np.array_equal(visible.numpy().view(np.uint32), per_step[-1].view(np.uint32))  # the per-step path's bits
    assert per_step[-1][0] != 0.0
    for k in ("obs", "rew", "reset", "progress", "successes"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    np.save(os.path.join(outdir, "cons%d.npy" % rank), visible.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["allegro", "shadow"])
def test_queued_shard_records_equal_the_per_step_exchange(tmp_path, which):
    world = 2
    mp.spawn(_queued_rank, args=(world, _free_port(), 1501, str(tmp_path), which), nprocs=world, join=True)
    cons = [np.load(os.path.join(str(tmp_path), "cons%d.npy" % r)) for r in range(world)]
    assert np.array_equal(cons[0].view(np.uint32), cons[1].view(np.uint32))
