#!/usr/bin/env python3
"""Lab: why is the first launch of a replay slower than the others?  Stamps after every launch of a 12-launch graph;
(a) one replay after an idle synchronisation, (b) the second of two back-to-back replays (no idle in between),
(c) one replay after an idle synchronisation that is preceded, in the same graph, by a throw-away launch of the same
kernel on a 4 096-lane handle (warms the code, not the data).  Run from the repository root on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, K = 1 << 20, 12
dev = torch.device("cuda", 0)
b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
acts = torch.randint(0, 5, (K, 2, N), dtype=torch.int8, device=dev)
obs = torch.empty((K, N), dtype=torch.int16, device=dev); rew = torch.empty((K, N), dtype=torch.int8, device=dev)
term = torch.empty((K, N), dtype=torch.uint8, device=dev); trunc = torch.empty((K, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); b.reset()
b.graph_begin(); b.stamp(2)
for k in range(K):
    b.step_plain(acts[k, 0], acts[k, 1], obs[k], rew[k], term[k], trunc[k]); b.stamp(3 + k)
g = b.graph_end()

def deltas():
    t, khz = b.stamps(2, K + 1)
    return np.diff(t.astype(np.int64)) / (khz * 1e-3)

for name, reps in (("(a) one replay after an idle sync", 1), ("(b) second of two back-to-back replays", 2), ("(b') fourth of four", 4)):
    rows = []
    for _ in range(5):
        torch.cuda.synchronize(); time.sleep(0.001)
        b.graph_launch(g, reps); torch.cuda.synchronize()
        rows.append(deltas())
    r = np.median(np.array(rows), axis=0)
    print("%-44s first %.2f  second %.2f  mean of the rest %.2f us" % (name, r[0], r[1], r[2:].mean()))
for idle_ms in (0.0, 0.02, 0.1, 1.0, 10.0):
    rows = []
    for _ in range(5):
        b.graph_launch(g, 1); torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        b.graph_launch(g, 1); torch.cuda.synchronize()
        rows.append(deltas())
    r = np.median(np.array(rows), axis=0)
    print("(c) replay %.2f ms after the previous one's sync: first %.2f  second %.2f  rest %.2f us" % (idle_ms, r[0], r[1], r[2:].mean()))

# (d) as bench.py has it: K = 20, several graphs alive, the stamped one replayed after OTHER graphs were
K2 = 20
acts2 = torch.randint(0, 5, (K2, 2, N), dtype=torch.int8, device=dev)
obs2 = torch.empty((K2, N), dtype=torch.int16, device=dev); rew2 = torch.empty((K2, N), dtype=torch.int8, device=dev)
term2 = torch.empty((K2, N), dtype=torch.uint8, device=dev); trunc2 = torch.empty((K2, N), dtype=torch.uint8, device=dev)
def enq(k): b.step_plain(acts2[k, 0], acts2[k, 1], obs2[k], rew2[k], term2[k], trunc2[k])
b.graph_begin()
for k in range(K2): enq(k)
g_plain = b.graph_end()
b.graph_begin(); b.timer_start()
for k in range(K2): enq(k)
b.timer_mark(); g_twin = b.graph_end()
b.graph_begin(); b.stamp(2)
for k in range(K2):
    enq(k); b.stamp(3 + k)
g_prof = b.graph_end()
def deltas2():
    t, khz = b.stamps(2, K2 + 1)
    return np.diff(t.astype(np.int64)) / (khz * 1e-3)
for name, pre in (("(d1) profile graph after itself", g_prof), ("(d2) profile graph after the plain graph", g_plain), ("(d3) profile graph after the twin", g_twin)):
    rows, twin = [], []
    for _ in range(5):
        b.graph_launch(pre, 1); torch.cuda.synchronize()
        b.graph_launch(g_prof, 1); torch.cuda.synchronize()
        rows.append(deltas2())
        b.graph_launch(g_twin, 1); twin.append(b.timer_read() * 1e3); torch.cuda.synchronize()
    r = np.median(np.array(rows), axis=0)
    print("%-44s first %.2f  second %.2f  rest %.2f us;  twin region %.1f us for %d launches" % (name, r[0], r[1], r[2:].mean(), np.median(twin), K2))
