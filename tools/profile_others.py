#!/usr/bin/env python3
"""Launches the kernels bench.py never times, 100 times each at 2^20 lanes, so that a `rocprofv3 --kernel-trace --stats` pass has
rows for them (tools/profile_run.sh, pass kt_other): batched_reset (all lanes / masked / with the observation out), the per-lane
fallback kernels (caller-supplied uniforms; a 13x9 pitch, beyond the byte arithmetic; max_steps = 200; misaligned buffers), the
single-agent step, and soccer_trajectory_returns.  Prints one JSON line with the launches made (no timing of its own)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SOCCER_HIP_RUNTIME", "system")
from gym_soccer_littman94_amd import SoccerBatch

N, REP = 1 << 20, 100
rng = np.random.default_rng(1)
made = {}


def bufs(b, n):
    return (b.alloc(n, np.int8).upload(rng.integers(0, 5, n, dtype=np.int8)), b.alloc(n, np.int8).upload(rng.integers(0, 5, n, dtype=np.int8)),
            b.alloc(n, np.uint16), b.alloc(n, np.int8), b.alloc(n, np.uint8), b.alloc(n, np.uint8))


b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False)
A, B, obs, rew, te, tr = bufs(b, N)
mask = b.alloc(N, np.uint8).upload((rng.random(N) < 0.3).astype(np.uint8))
for _ in range(REP):
    b.reset()
for _ in range(REP):
    b.reset(obs=obs)
for _ in range(REP):
    b.reset(mask=mask, obs=obs)
made["batched_reset: all lanes / + obs / masked + obs"] = 3 * REP
u = b.alloc(N, np.float64).upload(rng.random(N)); ur = b.alloc(N, np.float64).upload(rng.random(N))
for _ in range(REP):
    b.step(A, B, obs=obs, reward=rew, terminated=te, truncated=tr, u_step=u, u_reset=ur)
made["batched_step_ex with caller-supplied uniforms (slip 0)"] = REP
T = 64
R = b.alloc((T, N), np.int8).fill(0); TE = b.alloc((T, N), np.uint8).fill(0); TR = b.alloc((T, N), np.uint8).fill(0)
b.rollout(T, sample_actions=True, reward=R, terminated=TE, truncated=TR, out_stride=N)
last = b.alloc(N, np.int8); cnt = b.alloc(N, np.int32)
for _ in range(REP):
    b.trajectory_returns(T, R, TE, TR, N, last_return=last, episode_count=cnt, hist=False)
b.sync()
made["soccer_trajectory_returns, T = 64"] = REP
# misaligned result buffers: byte I/O
o1 = b.alloc(N + 8, np.uint8)
for _ in range(REP):
    b.step(A, B, obs=obs, reward=o1.ptr + 1, terminated=te, truncated=tr)
made["batched_step_ex, reward stream misaligned by 1 byte"] = REP
b.set_policy("player_b", rng.integers(0, 5, b.nS).astype(np.int8))
for _ in range(REP):
    b.step(A, None, obs=obs, reward=rew, terminated=te, truncated=tr)
made["single-agent batched_step (fixed player_b policy)"] = REP
b.sync(); b.close()

b = SoccerBatch(N, 5, 4, 0.2, seed=0, autoreset=True, step_stats=False)
A, B, obs, rew, te, tr = bufs(b, N)
u = b.alloc(N, np.float64).upload(rng.random(N)); ur = b.alloc(N, np.float64).upload(rng.random(N))
b.reset()
for _ in range(REP):
    b.step(A, B, obs=obs, reward=rew, terminated=te, truncated=tr, u_step=u, u_reset=ur)
made["batched_step_ex with caller-supplied uniforms (slip 0.2)"] = REP
b.sync(); b.close()

b = SoccerBatch(N, 13, 9, 0.0, seed=0, autoreset=True, step_stats=False)      # beyond the byte arithmetic (H * W > 128)
A, B, obs, rew, te, tr = bufs(b, N)
b.reset()
for _ in range(REP):
    b.step_plain(A, B, obs, rew, te, tr)
made["batched_step on a 13x9 pitch"] = REP
b.sync(); b.close()

b = SoccerBatch(N, 5, 4, 0.0, seed=0, autoreset=True, step_stats=False, max_steps=200)   # timestep beyond the guard bit
A, B, obs, rew, te, tr = bufs(b, N)
b.reset()
for _ in range(REP):
    b.step_plain(A, B, obs, rew, te, tr)
made["batched_step with max_steps = 200"] = REP
b.sync(); b.close()
print(json.dumps({"lanes": N, "launches": made}))
