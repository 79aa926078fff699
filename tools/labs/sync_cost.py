#!/usr/bin/env python3
"""What does closing a timed region cost on the host?  Run on the GPU box.  A 20-launch graph is replayed and its
closing stamp watched (soccer_timer_read), then the device is idle: how long do the candidates for the region's one
host synchronisation take from there?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, K = 1 << 20, 20
dev = torch.device("cuda", 0)
for own_stream in (True,):
    b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False, stream=None if own_stream else 0)
    acts = torch.randint(0, 5, (K, 2, N), dtype=torch.int8, device=dev)
    obs = torch.empty((K, N), dtype=torch.int16, device=dev); rew = torch.empty((K, N), dtype=torch.int8, device=dev)
    term = torch.empty((K, N), dtype=torch.uint8, device=dev); trunc = torch.empty((K, N), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    b.reset()
    b.graph_begin(); b.timer_start()
    for k in range(K):
        b.step_plain(acts[k, 0], acts[k, 1], obs[k], rew[k], term[k], trunc[k])
    b.timer_mark(); g = b.graph_end()
    b.graph_launch(g, 1); b.timer_read(); torch.cuda.synchronize()
    for name, fn in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("soccer_sync (hipStreamSynchronize)", b.sync),
                     ("torch.cuda.current_stream().synchronize()", lambda: torch.cuda.current_stream().synchronize()),
                     ("nothing", lambda: None)):
        ts, launch, wait = [], [], []
        for _ in range(30):
            torch.cuda.synchronize(); time.sleep(0.0005)
            t0 = time.perf_counter(); b.graph_launch(g, 1); t1 = time.perf_counter(); b.timer_read(); t2 = time.perf_counter(); fn(); t3 = time.perf_counter()
            launch.append(t1 - t0); wait.append(t2 - t1); ts.append(t3 - t2)
        print("%-12s %-44s launch call %.1f us, to stamp %.1f us, closing sync median %.1f us (min %.1f)" % (
            "own stream" if own_stream else "null stream", name, np.median(launch) * 1e6, np.median(wait) * 1e6, np.median(ts) * 1e6, min(ts) * 1e6))
    b.graph_destroy(g); b.close()

# the same question behind an (almost) empty replay: two stamp kernels and nothing else — no dirty lines to write back
b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
b.reset(); b.sync()
b.graph_begin(); b.timer_start(); b.timer_mark(); g = b.graph_end()
b.graph_launch(g, 1); b.timer_read(); torch.cuda.synchronize()
for name, fn in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("nothing", lambda: None)):
    ts, launch, wait = [], [], []
    for _ in range(30):
        torch.cuda.synchronize(); time.sleep(0.0005)
        t0 = time.perf_counter(); b.graph_launch(g, 1); t1 = time.perf_counter(); b.timer_read(); t2 = time.perf_counter(); fn(); t3 = time.perf_counter()
        launch.append(t1 - t0); wait.append(t2 - t1); ts.append(t3 - t2)
    print("%-12s %-44s launch call %.1f us, to stamp (+ marker) %.1f us, closing sync median %.1f us" % (
        "empty graph", name, np.median(launch) * 1e6, np.median(wait) * 1e6, np.median(ts) * 1e6))

# does it matter what the LAST packet of the stream is?  the same 20-launch replay, then one more stamp enqueued eagerly
# (a plain kernel launch carries its own completion signal), watched, then the synchronisation
b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
b.reset(); b.sync()
b.graph_begin(); b.timer_start()
for k in range(K):
    b.step_plain(acts[k, 0], acts[k, 1], obs[k], rew[k], term[k], trunc[k])
b.timer_mark(); g = b.graph_end()
b.graph_launch(g, 1); b.timer_read(); torch.cuda.synchronize()
for name, fn in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("soccer_sync", b.sync)):
    ts, launch, wait = [], [], []
    for _ in range(30):
        torch.cuda.synchronize(); time.sleep(0.0005); b.stamps_clear(5, 1)
        t0 = time.perf_counter(); b.graph_launch(g, 1); b.stamp(5); t1 = time.perf_counter()
        while int(b.stamps(5, 1)[0][0]) == 0:
            pass
        t2 = time.perf_counter(); fn(); t3 = time.perf_counter()
        launch.append(t1 - t0); wait.append(t2 - t1); ts.append(t3 - t2)
    print("%-12s %-44s launch calls %.1f us, to eager stamp %.1f us, closing sync median %.1f us (total %.1f)" % (
        "graph+eager", name, np.median(launch) * 1e6, np.median(wait) * 1e6, np.median(ts) * 1e6, np.median(np.array(launch) + np.array(wait) + np.array(ts)) * 1e6))
