"""Child process of tests/test_bench_launcher.py: what bench.py's rank processes do around the benchmark proper
(rendezvous from the environment the launcher set, one collective, rank 0 prints one JSON line)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if "--fail" in sys.argv and rank == 1:
    sys.exit(3)
if "--hang" in sys.argv:         # a rank stuck in a collective that never completes
    import time
    time.sleep(600)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": t.item(), "local": os.environ["LOCAL_RANK"],
                      "addr": os.environ["MASTER_ADDR"]}), flush=True)
dist.barrier()
dist.destroy_process_group()
