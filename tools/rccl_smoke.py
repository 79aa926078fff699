#!/usr/bin/env python3
"""One-rank rehearsal of the post-run exchange on a real GPU: torch.distributed backend "nccl" (= RCCL),
all_gather of int8 per-lane returns + all_reduce of the 3-bin histogram, exactly the calls bench.py --gpus N
makes after its timed region (the driver runs the N > 1 cases; one GPU only proves RCCL initialises here)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
import torch
import torch.distributed as dist
from gym_soccer_littman94_amd.distributed import gather_lane_values, reduce_histogram

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.randint(-1, 2, (1 << 20,), dtype=torch.int8, device=dev)
g = gather_lane_values(x, 1 << 20)
h = reduce_histogram([1, 2, 3], device=dev)
torch.cuda.synchronize()
assert torch.equal(g, x) and h == [1, 2, 3]
print("RCCL one-rank all_gather + all_reduce ok in %.1f s" % (time.time() - t0))
dist.destroy_process_group()
