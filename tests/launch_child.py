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
