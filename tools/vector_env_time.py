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
    K = 2000
    pairs = [{'player_a': ta[k % 8, 0], 'player_b': ta[k % 8, 1]} for k in range(K)]      # slicing is not the env's cost
    for k in range(50): v.step(pairs[k])
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(K): v.step(pairs[k])
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("device io  n=%8d: %8.2f us/step  %.3g env-steps/s (rewards as float32 / info['p'] left lazy)" % (n, dt / K * 1e6, n * K / dt))
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(K):
        o, r, te, tr, info = v.step(pairs[k]); r["player_a"]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("device io  n=%8d: %8.2f us/step  (reading the float32 reward of player_a every step: + one cast kernel)" % (n, dt / K * 1e6))
    # the floor under step(): the bare ctypes call it makes
    call, h, ref = v._step_call, v.batch.h, v._step_ref
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(K): call(h, ref)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("           n=%8d: %8.2f us per bare batched_step_ex ctypes call (same launch, no Python around it)" % (n, dt / K * 1e6))
    v.close()
