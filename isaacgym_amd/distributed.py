"""Multi-GPU layout of the step path: one process per GPU, environments sharded.

Every function on the path is per-environment, so a job of N environments on G
GPUs gives rank r the contiguous block `env_shard(N, r, G)` of every buffer and
the step itself moves no data between GPUs.  The only exchanges are tiny and
sit beside the path:

* AllegroHand's `consecutive_successes` averages over all environments
  (tasks/allegro_hand.py:714-719 of the reference): each shard leaves a 16-byte
  record, `gather_records` all-gathers them (RCCL over xGMI, one 16 x G byte
  collective) and every rank adds them in rank order, so all ranks hold the
  same value bit for bit;
* imitation batches (`fetch_amp_obs_demo`) are sampled per rank and
  `all_gather_rows` concatenates them when a global batch is wanted.

`torch.distributed` is used as shipped: backend "nccl" (= RCCL on ROCm) for
device tensors, "gloo" for host tensors (tests).
"""
import os

import torch
import torch.distributed as dist


def env_shard(num_envs, rank, world_size):
    """(start, count) of rank's contiguous block; the first num_envs % world_size ranks get one more."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(num_envs), int(world_size))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def rank_env_offset(num_envs_per_rank, rank=None):
    """First global environment index of a rank that builds `num_envs_per_rank` environments of its own
    (the reference's multi-GPU layout, utils/rlgames_utils.py:91-112): assign it to `task.env_offset`."""
    return (int(os.environ.get("RANK", "0")) if rank is None else int(rank)) * int(num_envs_per_rank)


def shard_seed(seed, rank):
    """A per-rank generator key, for shards that do not pass their `env_offset`.  The Philox counter is
    (environment index + env_offset): shards that set `env_offset = env_shard(...)[0]` and share ONE seed draw
    exactly what a single GPU would draw for the same environments; shards that leave env_offset at 0 need
    different keys, or they would all draw the same numbers."""
    return (int(seed) ^ (int(rank) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def init_from_env(backend=None):
    """Join the job described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (torch.distributed.run sets them).  Returns (rank, local_rank, world_size); a single
    process without those variables is a world of one and initialises nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


# A world of one needs no collective and the helpers below return at once.  Setting this makes them call the
# backend all the same (when a group is initialised): the only way to run the RCCL calls on a one-GPU machine.
FORCE_COLLECTIVES = False


def _alone(group=None):
    return world_size(group) == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _host_backend(group=None):
    """True when the group's backend moves host memory only (gloo has no all_gather_into_tensor for
    device tensors): device records then travel through the host."""
    return dist.get_backend(group) == "gloo"


def gather_records(record, group=None):
    """All-gather one fixed-size record per rank: 1-D tensor [R] -> [world, R], rank order."""
    record = record.contiguous()
    w = world_size(group)
    if _alone(group):
        return record.unsqueeze(0).clone()
    flat = record.reshape(-1)
    dev = record.device
    if dev.type != "cpu" and _host_backend(group):
        flat = flat.cpu()
    out = torch.empty(w * flat.numel(), dtype=record.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)  # concatenation along dim 0, rank order
    return out.to(dev).view((w,) + tuple(record.shape))


def all_gather_rows(rows, group=None):
    """Concatenate equally sized per-rank batches [n, ...] -> [world * n, ...], rank order."""
    rows = rows.contiguous()
    w = world_size(group)
    if _alone(group):
        return rows
    dev = rows.device
    if dev.type != "cpu" and _host_backend(group):
        rows = rows.cpu()
    out = torch.empty((w * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, rows, group=group)
    return out.to(dev)


def all_reduce_sum(t, group=None):
    """Sum an integer tensor over the ranks in place (counts of a histogram: the order of the additions cannot
    change the bits).  Device tensors travel through the host under a host-only backend."""
    if _alone(group):
        return t
    if t.device.type != "cpu" and _host_backend(group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def max_over_ranks(value, device, group=None):
    """MAX of a host scalar over the ranks (benchmark timing)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardRecordQueue:
    """Keeps a per-step exchange of small records off the step's critical path.

    A sharded task whose step ends in a tiny all-gather (AllegroHand / ShadowHand: the 16-byte
    {num_resets, sum} record behind `consecutive_successes`) would otherwise pay one collective latency
    per step on its own stream.  Here every step writes its record into a slot of a device-side queue
    (`slot()`, handed to the kernel as `shard_record`); once per `interval` steps `flush()` gathers the
    whole queue of every rank with ONE collective on a side stream and calls `apply(records, num_steps)`
    there, which advances the running value step by step in step order - the same arithmetic in the same
    order as the per-step exchange, so the bits are those of the per-step path.

    The running value lives in `pending` (owned by the side stream); `visible` - what the task exposes -
    is refreshed from it at the next flush, after the task stream has waited for the previous horizon's
    event.  So a reader on the task stream never races with the side stream and sees, at any step, the
    value as of the end of the last horizon but one.  `drain()` brings `visible` fully up to date
    (synchronises the task stream with the side stream)."""

    def __init__(self, visible, interval, group=None, record_words=2):
        self.visible = visible
        self.pending = visible.clone()
        self.interval = max(1, int(interval))
        self.group = group
        dev = visible.device
        self._cuda = dev.type == "cuda"
        self._rings = [torch.zeros(self.interval, record_words, dtype=torch.int64, device=dev) for _ in range(2)]
        # persistent views: the kernels' argument cache recognises a slot by the tensor object
        self._slots = [[r[i] for i in range(self.interval)] for r in self._rings]
        self._cur, self._count = 0, 0
        self._side = torch.cuda.Stream(device=dev) if self._cuda else None
        self._done = None          # event: the side stream finished the last flushed horizon
        self._dirty = False        # pending holds something visible has not seen
        self.flushes = 0

    def slot(self):
        """Where this step's record goes (a 16-byte tensor view)."""
        return self._slots[self._cur][self._count]

    def push(self):
        """The step that used slot() has been enqueued; True when the queue is full and must be flushed."""
        self._count += 1
        return self._count >= self.interval

    def _refresh_visible(self):
        if self._done is not None:
            torch.cuda.current_stream(self.visible.device).wait_event(self._done)
            self._done = None
        if self._dirty:
            self.visible.copy_(self.pending)
            self._dirty = False

    def flush(self, apply):
        """Gather the queued records of all ranks and run apply(records [world, steps * words], steps)."""
        k = self._count
        if k == 0:
            return
        ring = self._rings[self._cur][:k]
        self._cur ^= 1
        self._count = 0
        self._refresh_visible()  # the previous horizon's result becomes visible; its ring is free again
        if self._cuda:
            dev = self.visible.device
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                records = gather_records(ring.reshape(-1), self.group)
                apply(records.contiguous(), k)
                self._done = torch.cuda.Event()
                self._done.record(self._side)
        else:
            records = gather_records(ring.reshape(-1), self.group)
            apply(records.contiguous(), k)
        self._dirty = True
        self.flushes += 1

    def drain(self, apply):
        """Flush what is queued and make `visible` current."""
        self.flush(apply)
        self._refresh_visible()
