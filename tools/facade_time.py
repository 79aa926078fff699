import time, numpy as np, sys
sys.path.insert(0, '.')
import gym_soccer_littman94_amd as gsa
env = gsa.make("SoccerLittman94-v0")
acts = np.random.RandomState(123).randint(0, 5, size=(10000, 2))
env.reset(seed=0)
for rep in range(3):
    t = time.perf_counter(); n = 0
    for a, b in acts[:3000]:
        if env.needs_reset: env.reset()
        env.step({'player_a': int(a), 'player_b': int(b)}); n += 1
    dt = time.perf_counter() - t
    print("facade: %.1f us/step (%.3g steps/s)" % (dt / n * 1e6, n / dt))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for a, b in acts[:2000]:
    if env.needs_reset: env.reset()
    env.step({'player_a': int(a), 'player_b': int(b)})
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
