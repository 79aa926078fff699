#!/usr/bin/env python3
"""batched_reset at 2^20 lanes: all lanes / masked, us per call (HIP events over 200 calls)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from gym_soccer_littman94_amd import SoccerBatch
n = 1 << 20
b = SoccerBatch(n, 5, 4, 0.0, seed=0, autoreset=False)
obs = b.alloc(n, np.uint16)
mask = b.alloc(n, np.uint8).upload((np.arange(n) % 3 == 0).astype(np.uint8))
for tag, kw in (("all lanes, obs out", dict(obs=obs)), ("all lanes, no obs", {}), ("masked, obs out", dict(mask=mask, obs=obs))):
    for _ in range(10): b.reset(**kw)
    b.sync(); b.timer_start()
    for _ in range(200): b.reset(**kw)
    print("%-22s %.2f us per batched_reset" % (tag, b.timer_stop() * 1e3 / 200))
