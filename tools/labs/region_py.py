#!/usr/bin/env python3
"""The K = 20 timed region of bench.py from Python, by what closes it (run on the GPU box next to build/region_lab, which
does the same from C++): torch.cuda.synchronize(), hipDeviceSynchronize through ctypes, the handle's own stream."""
import ctypes, os, re, sys, time
sys.path.insert(0, ".")
import numpy as np
if os.environ.get("PRELOAD") == "1":      # libsoccer_hip.so first: its RUNPATH brings in /opt/rocm's HIP runtime, which torch then shares
    ctypes.CDLL(os.path.join("gym_soccer_littman94_amd", "libsoccer_hip.so"))
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, K = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so.7")
print("HIP runtime in this process:", sorted(set(re.findall(r"\S*libamdhip64\S*|\S*libhsa-runtime\S*", open("/proc/self/maps").read()))))
x = torch.arange(1 << 20, device=dev, dtype=torch.float32); assert float((x * 2).sum()) == float((1 << 20) * ((1 << 20) - 1))
b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
acts = torch.randint(0, 5, (K, 2, N), dtype=torch.int8, device=dev)
obs = torch.empty((K, N), dtype=torch.int16, device=dev); rew = torch.empty((K, N), dtype=torch.int8, device=dev)
term = torch.empty((K, N), dtype=torch.uint8, device=dev); trunc = torch.empty((K, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
b.reset()
b.graph_begin()
for k in range(K):
    b.step_plain(acts[k, 0], acts[k, 1], obs[k], rew[k], term[k], trunc[k])
g = b.graph_end()
for _ in range(3):
    b.graph_launch(g, 1); b.sync()
launch, handle, now = b.lib.soccer_graph_launch, b.h, time.perf_counter
hds = hip.hipDeviceSynchronize
ssync = b.lib.soccer_sync
for rep in range(2):
    for name, opener, closer in (("torch.cuda.synchronize", torch.cuda.synchronize, torch.cuda.synchronize),
                                 ("ctypes hipDeviceSynchronize", hds, hds),
                                 ("soccer_sync (hipStreamSynchronize)", hds, lambda: ssync(handle)),
                                 ("open torch / close hipDeviceSynchronize", torch.cuda.synchronize, hds)):
        wall, enq = [], []
        for _ in range(40):
            opener()
            t0 = now(); launch(handle, g, 1); t1 = now(); closer(); t2 = now()
            wall.append(t2 - t0); enq.append(t1 - t0)
        print("K=%d %-42s wall median %.1f us  min %.1f   launch call %.1f us" % (K, name, np.median(wall) * 1e6, min(wall) * 1e6, np.median(enq) * 1e6))
# eager head (and tail) around a shorter replay: does the replay's start-up hide behind the head's launches?
step = b.lib.batched_step
ptrs = [(acts[k, 0].data_ptr(), acts[k, 1].data_ptr(), obs[k].data_ptr(), rew[k].data_ptr(), term[k].data_ptr(), trunc[k].data_ptr(), None) for k in range(K)]
for head, tail in ((2, 0), (4, 0), (0, 2), (2, 2)):
    b.graph_begin()
    for k in range(head, K - tail):
        b.step_plain(acts[k, 0], acts[k, 1], obs[k], rew[k], term[k], trunc[k])
    g2 = b.graph_end()
    for rep in range(2):
        wall, enq = [], []
        for it in range(43):
            hds()
            t0 = now()
            for a in ptrs[:head]:
                step(handle, *a)
            launch(handle, g2, 1)
            for a in ptrs[K - tail:]:
                step(handle, *a)
            t1 = now(); hds(); t2 = now()
            if it >= 3:
                wall.append(t2 - t0); enq.append(t1 - t0)
        print("K=%d head %d + graph + tail %d   wall median %.1f us  min %.1f   enqueue %.1f us" % (K, head, tail, np.median(wall) * 1e6, min(wall) * 1e6, np.median(enq) * 1e6))
    b.graph_destroy(g2)
b.graph_destroy(g); b.close()
