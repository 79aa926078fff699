#!/usr/bin/env python3
"""Fused rollout (T = 100, 2^20 lanes, four trajectories out) when every launch reads action rows nobody has touched for a long
time (R distinct [T, 2, N] blocks, 200 MB each, visited round robin: streaming from HBM) against launches that re-read one
block.  Run once per library variant (tools/labs/lib_ab-style: copy build/lib_<v>.so over the package's library first)."""
import sys, time
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, T, R = 1 << 20, 100, 6
dev = torch.device("cuda", 0)
b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
acts = torch.randint(0, 5, (R, T, 2, N), dtype=torch.int8, device=dev)
obs = torch.empty((T, N), dtype=torch.int16, device=dev); rew = torch.empty((T, N), dtype=torch.int8, device=dev)
term = torch.empty((T, N), dtype=torch.uint8, device=dev); trunc = torch.empty((T, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); b.reset()
def run(r):
    b.rollout(T, act_a=acts[r, 0, 0], act_b=acts[r, 0, 1], act_stride=2 * N, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=N)
for name, order in (("one block re-read", [0] * 12), ("six blocks round robin", list(range(R)) * 2)):
    for r in order[:3]: run(r)
    b.sync(); t0 = time.perf_counter()
    for r in order: run(r)
    b.sync(); dt = time.perf_counter() - t0
    print("%-24s %.3f us per step  %.3g env-steps/s" % (name, dt / (len(order) * T) * 1e6, N * T * len(order) / dt))
b.close()
