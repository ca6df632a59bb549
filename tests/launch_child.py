"""Child process of tests/test_launch.py: behaves as the test's arguments say."""
import json
import os
import sys
import time

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "ok":
    if rank != 0:
        print("noise from rank %d" % rank)  # must not reach the relayed output
    else:
        print(json.dumps({"n_gpus": world, "local_rank": int(os.environ["LOCAL_RANK"]),
                          "addr": os.environ["MASTER_ADDR"], "port": int(os.environ["MASTER_PORT"])}))
elif mode == "fail":  # the last rank fails, the others would run for a minute
    if rank == world - 1:
        sys.exit(7)
    time.sleep(60)
elif mode == "hang":
    time.sleep(60)
elif mode == "gloo":  # a real rendezvous over the variables the launcher set
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"sum": float(t.item())}))
    dist.destroy_process_group()
elif mode in ("bench_group", "bench_group_fallback"):
    # bench.py's group bring-up with RCCL asked for on a machine where it cannot come up (no GPU): every rank
    # must agree on the failure and leave with code 3, or, with the fallback allowed, carry on over gloo
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    g = bench.init_group(dist, torch, "nccl", torch.device("cpu"), mode == "bench_group_fallback", timeout_s=60.0)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, group=g.data)
    if rank == 0:
        print(json.dumps({"backend": g.backend, "rccl_ranks": g.rccl_ranks, "note": g.note, "sum": float(t.item())}))
    dist.barrier(group=g.control)
    dist.destroy_process_group()
