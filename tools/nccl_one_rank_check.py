"""RCCL code paths of the package on ONE rank (all a one-GPU box offers): the collectives run through the nccl
backend on device tensors - no host staging as under gloo - for the dtypes and shapes the product uses.
Usage: PYTHONPATH=. python tools/nccl_one_rank_check.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = torch.empty(32, dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(out, torch.arange(32, dtype=torch.uint8, device=dev))
torch.cuda.synchronize()
print("nccl ok", float(t), int(out.sum()))

from isaacgym_amd import _abi, distributed, ops, synthetic  # noqa: E402

distributed.FORCE_COLLECTIVES = True  # a world of one would otherwise skip the backend

# the global quantile in two collectives: int32 all-reduce of the histogram, int64 all-gather of the slots
x = torch.randn(1 << 20, dtype=torch.float64, device=dev).abs()
wq = ops.WindowQuantile(dev)
ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=dev)
for step in range(3):
    assert wq.run(x, 0.99, x.numel(), distributed.all_reduce_sum, distributed.gather_records)
    assert float(wq.result()) == float(torch.quantile(x.cpu(), 0.99))
assert float(ops.quantile_sharded(x, 0.99, x.numel(), ws, distributed.all_reduce_sum)) == float(torch.quantile(x.cpu(), 0.99))
print("window quantile over nccl ok, collectives per call:", wq.collectives)

# the hands' queued shard records: one all-gather per interval on the side stream
n = 10007
p = synthetic.allegro_params("full_state", seed=1, rng_offset=2)
full = synthetic.allegro_state(n, 8, dev, "full_state")
twin = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
q = distributed.ShardRecordQueue(full["consecutive_successes"], 2)
apply = lambda records, k: ops.allegro_finalize_shards(p, dict(consecutive_successes=q.pending), records, k)
for step in range(4):
    for d in (full, twin):
        d["reset"][step::7] = 1
    full["shard_record"] = q.slot()
    ops.allegro_post_physics_shard(p, full, n)
    if q.push():
        q.flush(apply)
    ops.allegro_post_physics(p, twin, n)
q.drain(apply)
torch.cuda.synchronize()
a, b = float(full["consecutive_successes"]), float(twin["consecutive_successes"])
assert abs(a - b) <= 1e-6 * abs(b), (a, b)
print("queued shard records over nccl ok", a, "flushes", q.flushes)
dist.barrier()
dist.destroy_process_group()
