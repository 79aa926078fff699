#!/usr/bin/env python3
"""Where does a fused rollout step spend its time?  T = 100 at 2^20 lanes with and without the trajectory stores /
the action streams.  Run on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from gym_soccer_littman94_amd import SoccerBatch

N, T = 1 << 20, 100
slip = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda", 0)
b = SoccerBatch(N, 5, 4, slip, seed=0, autoreset=True)
b.reset()
acts = torch.randint(0, 5, (T, 2, N), dtype=torch.int8, device=dev)
obs = torch.empty((T, N), dtype=torch.int16, device=dev); rew = torch.empty((T, N), dtype=torch.int8, device=dev)
term = torch.empty((T, N), dtype=torch.uint8, device=dev); trunc = torch.empty((T, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()

def run(tag, **kw):
    b.rollout(T, **kw); b.sync()
    reps = []
    for _ in range(5):
        b.timer_start(); b.rollout(T, **kw); reps.append(b.timer_stop())
    ms = sorted(reps)[2]
    print("%-46s %.3f us/step  %.4g env-steps/s" % (tag, ms * 1e3 / T, N * T / (ms * 1e-3)))

A = dict(act_a=acts[0, 0], act_b=acts[0, 1], act_stride=2 * N)
O = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=N)
run("streams in, 4 trajectories out (bench)", **A, **O)
run("streams in, nothing out", **A)
run("streams in, obs only", **A, obs=obs, out_stride=N)
run("streams in, reward+term+trunc only", **A, reward=rew, terminated=term, truncated=trunc, out_stride=N)
run("sampled uniform, 4 trajectories out", sample_actions=True, **O)
run("sampled uniform, nothing out", sample_actions=True)

# BASELINE config 5: both players sample from [nS, 5] mixed-policy tables in the kernel
rngp = np.random.default_rng(94)
ta = SoccerBatch.mixed_policy_thresholds(rngp.dirichlet(np.ones(5) * 0.7, size=b.nS))
tb = SoccerBatch.mixed_policy_thresholds(rngp.dirichlet(np.ones(5) * 0.7, size=b.nS))
da = torch.from_numpy(ta.view(np.int16)).to(dev); db = torch.from_numpy(tb.view(np.int16)).to(dev)
run("config 5: mixed-policy tables, nothing out", sample_actions=True, mix_a=da, mix_b=db)
run("config 5: mixed-policy tables, 4 trajectories out", sample_actions=True, mix_a=da, mix_b=db, **O)
