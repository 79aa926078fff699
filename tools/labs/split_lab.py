#!/usr/bin/env python3
"""Lab: do independent lane shards on separate streams hide each other's kernel boundaries?  K captured steps of 2^20
lanes as ONE chain (one handle), and as S chains of 2^20 / S lanes each (S handles, own stream and graph each, replayed
together).  Device time = first opening stamp to last closing stamp (all on the device's one wall clock).  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, K = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
acts = torch.randint(0, 5, (K, 2, N), dtype=torch.int8, device=dev)
obs = torch.empty((K, N), dtype=torch.int16, device=dev); rew = torch.empty((K, N), dtype=torch.int8, device=dev)
term = torch.empty((K, N), dtype=torch.uint8, device=dev); trunc = torch.empty((K, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for S in (1, 2, 4, 8):
    n = N // S
    hs = [SoccerBatch(n, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False, lane_offset=i * n) for i in range(S)]
    gs = []
    for i, b in enumerate(hs):
        b.reset(); b.sync()
        sl = slice(i * n, (i + 1) * n)
        b.graph_begin(); b.timer_start()
        for k in range(K):
            b.step_plain(acts[k, 0, sl], acts[k, 1, sl], obs[k, sl], rew[k, sl], term[k, sl], trunc[k, sl])
        b.timer_mark(); gs.append(b.graph_end())
    res, walls = [], []
    for rep in range(12):
        torch.cuda.synchronize(); time.sleep(0.0005)
        t0 = time.perf_counter()
        for b, g in zip(hs, gs):
            b.graph_launch(g, 1)
        for b in hs:
            b.timer_read()
        t1 = time.perf_counter()
        st = [b.stamps(0, 2) for b in hs]
        khz = st[0][1]
        start = min(int(s[0][0]) for s in st); end = max(int(s[0][1]) for s in st)
        res.append((end - start) / (khz * 1e-3)); walls.append((t1 - t0) * 1e6)
    print("%d chain(s) of %7d lanes: device %.1f us for %d steps = %.3f us per step of 2^20 lanes (min %.3f); host wall %.1f us"
          % (S, n, np.median(res), K, np.median(res) / K, min(res) / K, np.median(walls)))
    for b, g in zip(hs, gs):
        b.graph_destroy(g); b.close()
