#!/usr/bin/env python3
"""Does it matter whether consecutive steps write their results to the same buffers (what an RL loop does) or to the rows of
a [K, N] trajectory (what bench.py does)?  K graph-replayed launches, device clock around them."""
import sys
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import SoccerBatch

n, K = 1 << 20, 300
dev = torch.device("cuda", 0)
for rep in range(2):
    for what in ("rows of a [K, N] trajectory, K action rows", "the same result buffers, K action rows", "the same result buffers, 8 action rows",
                 "rows of a [K, N] trajectory, 8 action rows"):
        b = SoccerBatch(n, 5, 4, 0.0, seed=1, autoreset=True, step_stats=False)
        same_out = what.startswith("the same"); ka = 8 if "8 action" in what else K
        acts = torch.randint(0, 5, (ka, 2, n), dtype=torch.int8, device=dev)
        ko = 1 if same_out else K
        obs = torch.empty((ko, n), dtype=torch.int16, device=dev); rew = torch.empty((ko, n), dtype=torch.int8, device=dev)
        term = torch.empty((ko, n), dtype=torch.uint8, device=dev); trunc = torch.empty((ko, n), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        b.reset()
        b.graph_begin(); b.timer_start()
        for k in range(K):
            o = 0 if same_out else k
            b.step_plain(acts[k % ka, 0], acts[k % ka, 1], obs[o], rew[o], term[o], trunc[o])
        b.timer_mark(); g = b.graph_end()
        ms = []
        for _ in range(5):
            b.graph_launch(g, 1); ms.append(b.timer_read())
        print("%-48s %.3f us per launch (median of 5 replays; first %.3f)" % (what, sorted(ms)[2] * 1e3 / K, ms[0] * 1e3 / K))
        b.graph_destroy(g); b.close(); del acts, obs, rew, term, trunc
