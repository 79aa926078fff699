#!/usr/bin/env python3
"""VectorSoccerEnv(io="device") enqueues on torch's CURRENT stream.  Does it matter for the host-bound step() whether that is the
legacy default (null) stream or a stream created by torch.cuda.Stream()?"""
import sys, time
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import VectorSoccerEnv

N, KV = 1 << 20, 1000
dev = torch.device("cuda", 0)
acts = torch.randint(0, 5, (20, 2, N), dtype=torch.int8, device=dev)
pairs = [{"player_a": acts[k % 20, 0], "player_b": acts[k % 20, 1]} for k in range(KV)]
side = torch.cuda.Stream(device=dev)
for rep in range(2):
    for name, ctx in (("default stream", torch.cuda.stream(torch.cuda.default_stream(dev))), ("torch.cuda.Stream()", torch.cuda.stream(side))):
        for info in (False, True):
            with ctx:
                v = VectorSoccerEnv(N, seed=0, io="device", info=info)
                v.reset()
                for k in range(20): v.step(pairs[k])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(KV): v.step(pairs[k])
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                v.close()
            print("%-22s info=%-5s %.2f us per step()" % (name, info, dt / KV * 1e6))
