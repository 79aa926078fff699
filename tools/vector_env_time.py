#!/usr/bin/env python3
"""Throughput of the gym-style VectorSoccerEnv in both I/O modes (host arrays over PCIe vs device tensors)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import VectorSoccerEnv

sizes = [int(x) for x in sys.argv[1:]] or [1 << 16, 1 << 20]
for n in sizes:
    rng = np.random.default_rng(0)
    a = rng.integers(0, 5, size=(8, 2, n)).astype(np.int8)
    for copy in (True, False):
        v = VectorSoccerEnv(n, seed=0, copy=copy)
        v.reset()
        for k in range(3): v.step({'player_a': a[k, 0], 'player_b': a[k, 1]})
        t = time.perf_counter(); K = 30 if n > 65536 else 2000
        for k in range(K): v.step({'player_a': a[k % 8, 0], 'player_b': a[k % 8, 1]})
        dt = time.perf_counter() - t
        print("numpy  io  n=%8d copy=%-5s: %8.1f us/step  %.3g env-steps/s" % (n, copy, dt / K * 1e6, n * K / dt))
        v.close()
    v = VectorSoccerEnv(n, seed=0, io="device")
    v.reset()
    ta = torch.from_numpy(a).cuda()
    for k in range(3): v.step({'player_a': ta[k, 0], 'player_b': ta[k, 1]})
    torch.cuda.synchronize(); t = time.perf_counter(); K = 300
    for k in range(K): v.step({'player_a': ta[k % 8, 0], 'player_b': ta[k % 8, 1]})
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("device io  n=%8d: %8.1f us/step  %.3g env-steps/s (incl. the wrapper's torch ops for rewards/flags/info)" % (n, dt / K * 1e6, n * K / dt))
    v.close()
