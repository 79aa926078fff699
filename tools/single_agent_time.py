#!/usr/bin/env python3
"""batched_step throughput in single-agent mode (one side plays a fixed policy looked up by the kernel)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import SoccerBatch

n, K = 1 << 20, 300
dev = torch.device("cuda", 0)
acts = torch.randint(0, 5, (8, n), dtype=torch.int8, device=dev)
for slip, fixed in ((0.0, True), (0.0, False), (0.2, True), (0.2, False)):
    b = SoccerBatch(n, 5, 4, slip, seed=1, autoreset=True, step_stats=False)
    if fixed:
        b.set_policy("player_b", np.random.default_rng(0).integers(0, 5, b.nS).astype(np.int8))
    obs = torch.empty(n, dtype=torch.int16, device=dev); rew = torch.empty(n, dtype=torch.int8, device=dev)
    term = torch.empty(n, dtype=torch.uint8, device=dev); trunc = torch.empty(n, dtype=torch.uint8, device=dev)
    b.reset()
    other = (lambda k: None) if fixed else (lambda k: acts[(k + 3) % 8])
    for k in range(20): b.step_plain(acts[k % 8], other(k), obs, rew, term, trunc)
    b.graph_begin()
    for k in range(K): b.step_plain(acts[k % 8], other(k), obs, rew, term, trunc)
    g = b.graph_end()
    b.sync(); b.timer_start(); b.graph_launch(g, 1); ms = b.timer_stop()
    print("%s batched_step slip %.1f: %.2f us/launch, %.3g env-steps/s" % ("single-agent" if fixed else "two-agent (same harness)", slip, ms * 1e3 / K, n * K / (ms * 1e-3)))
    b.close()
