#!/usr/bin/env python3
"""Where does the host's time per eager step go?  Run on the GPU box.  2^20 lanes are device-bound at ~4.2 us per step,
so the host-side cost is measured on a 4 096-lane handle (kernel ~2 us, the queue never fills): wall per call of
  (a) the bare ctypes call of batched_step with prebuilt arguments,
  (b) SoccerBatch.step_plain (argument conversion in Python),
  (c) VectorSoccerEnv(io="device", info=False).step (the gym-style surface)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from gym_soccer_littman94_amd import SoccerBatch, VectorSoccerEnv

dev = torch.device("cuda", 0)
for n in (4096, 1 << 20):
    b = SoccerBatch(n, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
    a = torch.randint(0, 5, (2, n), dtype=torch.int8, device=dev)
    obs = torch.empty(n, dtype=torch.int16, device=dev); rew = torch.empty(n, dtype=torch.int8, device=dev)
    term = torch.empty(n, dtype=torch.uint8, device=dev); trunc = torch.empty(n, dtype=torch.uint8, device=dev)
    b.reset(); torch.cuda.synchronize()
    K = 2000
    args = (a[0].data_ptr(), a[1].data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(), None)
    step, h = b.lib.batched_step, b.h
    for name, fn in (("bare ctypes batched_step", lambda: step(h, *args)),
                     ("SoccerBatch.step_plain", lambda: b.step_plain(a[0], a[1], obs, rew, term, trunc))):
        for _ in range(200): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("n %8d  %-34s enqueue %.2f us per call, drained after another %.1f us" % (n, name, (t1 - t0) / K * 1e6, (t2 - t1) * 1e6))
    b.close()
    for info in (True, False):
        v = VectorSoccerEnv(n, seed=0, io="device", info=info); v.reset()
        act = {"player_a": a[0], "player_b": a[1]}
        for _ in range(200): v.step(act)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): v.step(act)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("n %8d  %-34s enqueue %.2f us per call, drained after another %.1f us" % (n, "VectorSoccerEnv.step info=%s" % info, (t1 - t0) / K * 1e6, (t2 - t1) * 1e6))
        v.close()
