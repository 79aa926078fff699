#!/usr/bin/env python3
"""One handle ABOVE the byte-parallel kernels' 2^30-lane launch limit, for real (no SOCCER_SWAR_LAUNCH_LANES override): 2^30 + 2^20 + 4
lanes (~14 GB of device memory) — batched_reset, three batched_step calls, a 5-step batched_rollout and soccer_trajectory_returns — compared
slice by slice (the first lanes, the lanes around the 2^30 boundary, the last lanes) with small handles that own exactly those global lanes
(lane_offset, same seed, same actions): results depend on (seed, global lane, tick) only, so they must be identical.  GPU box; ~1 minute."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SOCCER_HIP_RUNTIME", "system")
from gym_soccer_littman94_amd import SoccerBatch

N = (1 << 30) + (1 << 20) + 4
W = 1 << 15
slices = [(0, 2 * W), ((1 << 30) - W, (1 << 30) + W), (N - 2 * W, N)]
seed, slip, T = 1234, float(sys.argv[1]) if len(sys.argv) > 1 else 0.0, 5
t0 = time.time()
rng = np.random.default_rng(5)
big = SoccerBatch(N, 5, 4, slip, seed=seed, autoreset=True, step_stats=False)
small = [SoccerBatch(hi - lo, 5, 4, slip, seed=seed, autoreset=True, lane_offset=lo, step_stats=False) for lo, hi in slices]
K = 3 + T
# actions: random in the compared slices, a constant elsewhere (2 GB per stream would be slow to draw; the slices are what is compared)
acts = [[rng.integers(0, 5, size=(2, hi - lo), dtype=np.int8) for lo, hi in slices] for _ in range(K)]
A = big.alloc((K, 2, N), np.int8).fill(3)
for k in range(K):
    for (lo, hi), a in zip(slices, acts[k]):
        for pl in range(2):
            big._check(big.lib.soccer_memcpy_h2d(big.h, A.ptr + (k * 2 + pl) * N + lo, a[pl].ctypes.data, hi - lo))
obs = big.alloc(N, np.uint16); rew = big.alloc(N, np.int8); te = big.alloc(N, np.uint8); tr = big.alloc(N, np.uint8)


def part(arr, lo, hi, dtype):
    out = np.empty(hi - lo, dtype)
    big._check(big.lib.soccer_memcpy_d2h(big.h, out.ctypes.data, arr.ptr + lo * np.dtype(dtype).itemsize, out.nbytes))
    return out


big.reset(obs=obs)
so = []
for s, (lo, hi) in zip(small, slices):
    o = s.alloc(hi - lo, np.uint16); s.reset(obs=o); so.append(o)
    assert np.array_equal(part(obs, lo, hi, np.uint16), o.download()), "reset, lanes %d..%d" % (lo, hi)
for k in range(3):
    big.step_plain(A.ptr + (2 * k) * N, A.ptr + (2 * k + 1) * N, obs, rew, te, tr)
    for s, (lo, hi), a, o in zip(small, slices, acts[k], so):
        n = hi - lo
        da = s.alloc(n, np.int8).upload(a[0]); db = s.alloc(n, np.int8).upload(a[1])
        r = s.alloc(n, np.int8); e = s.alloc(n, np.uint8); u = s.alloc(n, np.uint8)
        s.step_plain(da, db, o, r, e, u)
        for name, arr, ref, dt in (("obs", obs, o, np.uint16), ("reward", rew, r, np.int8), ("terminated", te, e, np.uint8), ("truncated", tr, u, np.uint8)):
            assert np.array_equal(part(arr, lo, hi, dt), ref.download()), "step %d %s, lanes %d..%d" % (k, name, lo, hi)
# fused rollout with trajectories + the reduction over them
R = big.alloc((T, N), np.int8); TE = big.alloc((T, N), np.uint8); TR = big.alloc((T, N), np.uint8)
big.rollout(T, A.ptr + 6 * N, A.ptr + 7 * N, act_stride=2 * N, reward=R, terminated=TE, truncated=TR, out_stride=N)
last = big.alloc(N, np.int8); cnt = big.alloc(N, np.int32)
hist = big.trajectory_returns(T, R, TE, TR, N, last_return=last, episode_count=cnt)
for s, (lo, hi), in zip(small, slices):
    n = hi - lo
    a = np.stack([np.stack([acts[3 + j][slices.index((lo, hi))][pl] for pl in range(2)]) for j in range(T)])       # [T, 2, n]
    da = s.alloc((T, 2, n), np.int8).upload(a)
    r = s.alloc((T, n), np.int8); e = s.alloc((T, n), np.uint8); u = s.alloc((T, n), np.uint8)
    s.rollout(T, da.ptr, da.ptr + n, act_stride=2 * n, reward=r, terminated=e, truncated=u, out_stride=n)
    l2 = s.alloc(n, np.int8); c2 = s.alloc(n, np.int32)
    s.trajectory_returns(T, r, e, u, n, last_return=l2, episode_count=c2)
    rr = r.download()
    for j in range(T):
        got = np.empty(n, np.int8)
        big._check(big.lib.soccer_memcpy_d2h(big.h, got.ctypes.data, R.ptr + j * N + lo, n))
        assert np.array_equal(got, rr[j]), "rollout step %d reward, lanes %d..%d" % (j, lo, hi)
    assert np.array_equal(part(last, lo, hi, np.int8), l2.download()) and np.array_equal(part(cnt, lo, hi, np.int32), c2.download()), "returns %d..%d" % (lo, hi)
assert int(hist.sum()) >= 0 and big.misuse() == 0 and big.tick == 1 + 3 + T
for s in small:
    assert s.tick == big.tick
    s.close()
big.close()
print("huge handle ok: %d lanes (two launches per call), slip %g: reset, 3 steps, a %d-step rollout and the trajectory reduction identical to "
      "small handles on the lanes %s in %.0f s" % (N, slip, T, slices, time.time() - t0))
