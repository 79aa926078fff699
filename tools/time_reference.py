#!/usr/bin/env python3
"""BASELINE config 1: time the REAL reference's step() loop in the build container (it cannot travel to the
GPU box).  Uses the same throw-away gym stand-in as tests/golden/make_golden.py.
Usage: python tools/time_reference.py [slip_prob]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import make_golden
make_golden._install_gym_stand_in()
from gym_soccer.envs import SoccerSimultaneousEnv

slip = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
env = SoccerSimultaneousEnv(width=5, height=4, slip_prob=slip)
acts = np.random.RandomState(123).randint(0, 5, size=(10000, 2))
best = 0.0
for rep in range(5):
    env.reset(seed=0)
    t = time.perf_counter()
    for a, b in acts:
        if env.needs_reset:
            env.reset()
        env.step({'player_a': a, 'player_b': b})
    dt = time.perf_counter() - t
    best = max(best, len(acts) / dt)
print("reference step() loop, slip_prob=%g: best of 5 = %.3g env-steps/s (%.1f us/step), 1 core" % (slip, best, 1e6 / best))
