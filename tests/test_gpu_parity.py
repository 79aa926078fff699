"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bar: bit-exact — every lane, every step: state, observation, reward, terminated, truncated,
probability code (integer / byte work; the only floating point is the float64 running-sum
comparison of the slip lists, which must select the same index as the reference does).
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gym_soccer_littman94_amd import SoccerBatch
from oracle.oracle import Oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TABLES = sorted(glob.glob(os.path.join(GOLDEN, "table_*.npz")))
REPLAYS = sorted(glob.glob(os.path.join(GOLDEN, "replay_*.npz")))
RESETS = sorted(glob.glob(os.path.join(GOLDEN, "reset_*.npz")))
TRAJS = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def _ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


class Bufs:
    """Device I/O buffers for one SoccerBatch."""
    def __init__(self, b, with_u=False):
        n = b.n
        self.b = b
        self.act_a = b.alloc(n, np.int8); self.act_b = b.alloc(n, np.int8)
        self.obs = b.alloc(n, np.uint16); self.reward = b.alloc(n, np.int8)
        self.term = b.alloc(n, np.uint8); self.trunc = b.alloc(n, np.uint8)
        self.code = b.alloc(n, np.uint8); self.fin = b.alloc(n, np.uint16)
        self.last = b.alloc(n, np.int8).fill(0)
        self.u_step = b.alloc(n, np.float64) if with_u else None
        self.u_reset = b.alloc(n, np.float64) if with_u else None

    def step(self, a, bb, u_step=None, u_reset=None):
        self.act_a.upload(a); self.act_b.upload(bb)
        if u_step is not None: self.u_step.upload(u_step)
        if u_reset is not None: self.u_reset.upload(u_reset)
        self.b.step(self.act_a, self.act_b, obs=self.obs, reward=self.reward, terminated=self.term,
                    truncated=self.trunc, prob_code=self.code, final_obs=self.fin, last_return=self.last,
                    u_step=self.u_step if u_step is not None else None,
                    u_reset=self.u_reset if u_reset is not None else None)
        return dict(obs=self.obs.download(), reward=self.reward.download(), terminated=self.term.download(),
                    truncated=self.trunc.download(), prob_code=self.code.download(), final_obs=self.fin.download())


def assert_state_equal(b, o):
    s = b.get_state()
    np.testing.assert_array_equal(s["row_a"], o.row_a); np.testing.assert_array_equal(s["col_a"], o.col_a)
    np.testing.assert_array_equal(s["row_b"], o.row_b); np.testing.assert_array_equal(s["col_b"], o.col_b)
    np.testing.assert_array_equal(s["poss"], o.poss & 1)
    np.testing.assert_array_equal(s["needs_reset"], (o.poss >> 1) & 1)
    np.testing.assert_array_equal(s["t"], o.t)


def assert_out_equal(g, c):
    for k in ("obs", "reward", "terminated", "truncated", "prob_code", "final_obs"):
        np.testing.assert_array_equal(g[k], c[k], err_msg=k)


@pytest.mark.parametrize("path", TABLES, ids=_ids(TABLES))
def test_rule_tables_match_reference(path):
    g = np.load(path)
    b = SoccerBatch(4, int(g["width"]), int(g["height"]), float(g["slip"]))
    lut, gv, isd = b.tables()
    assert b.nS == int(g["nS"])
    np.testing.assert_array_equal(lut, g["lut"])
    np.testing.assert_array_equal(gv, g["goal_value"])
    np.testing.assert_array_equal(isd, g["isd_states"])
    # slip weights * outcome probabilities exactly as the reference computes them (float64)
    probs = set(np.unique(g["prob"]).tolist())
    assert probs <= set(b.prob_table.tolist()), (probs, b.prob_table)
    b.close()


@pytest.mark.parametrize("epw", [1, 4, 8])
@pytest.mark.parametrize("path", REPLAYS, ids=_ids(REPLAYS))
def test_replay_vectors_from_reference_step(path, epw):
    """(state, t, joint action, u) vectors that were pushed through the REAL reference step()."""
    g = np.load(path)
    n = len(g["u"])
    b = SoccerBatch(n, int(g["width"]), int(g["height"]), float(g["slip"]), envs_per_thread=epw)
    st = g["state"]
    b.set_state(st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], t=g["t"], needs_reset=0)
    io = Bufs(b, with_u=True)
    out = io.step(g["action"][:, 0], g["action"][:, 1], u_step=g["u"], u_reset=np.zeros(n))
    s = b.get_state(); ns = g["next_state"]
    np.testing.assert_array_equal(s["row_a"], ns[:, 0]); np.testing.assert_array_equal(s["col_a"], ns[:, 1])
    np.testing.assert_array_equal(s["row_b"], ns[:, 2]); np.testing.assert_array_equal(s["col_b"], ns[:, 3])
    np.testing.assert_array_equal(s["poss"], ns[:, 4])
    np.testing.assert_array_equal(s["needs_reset"], g["needs_reset"])
    np.testing.assert_array_equal(out["obs"], g["obs"])
    np.testing.assert_array_equal(out["reward"].astype(np.float64), g["reward_a"])
    np.testing.assert_array_equal(out["terminated"], g["terminated"])
    np.testing.assert_array_equal(out["truncated"], g["truncated"])
    np.testing.assert_array_equal(np.round(b.prob_table[out["prob_code"]], 2), g["p"])
    assert b.stats()[1] == 0
    b.close()


@pytest.mark.parametrize("path", RESETS, ids=_ids(RESETS))
def test_reset_vectors_from_reference_reset(path):
    g = np.load(path)
    n = len(g["u"])
    b = SoccerBatch(n, int(g["width"]), int(g["height"]), float(g["slip"]) if "slip" in g.files else 0.0)
    u = b.alloc(n, np.float64).upload(g["u"]); obs = b.alloc(n, np.uint16)
    b.reset(u_reset=u, obs=obs)
    s = b.get_state(); st = g["state"]
    np.testing.assert_array_equal(obs.download(), g["obs"])
    for k, c in (("row_a", 0), ("col_a", 1), ("row_b", 2), ("col_b", 3), ("poss", 4)):
        np.testing.assert_array_equal(s[k], st[:, c])
    assert not s["t"].any() and not s["needs_reset"].any()
    b.close()


@pytest.mark.parametrize("path", TRAJS, ids=_ids(TRAJS))
def test_mt19937_trajectories_lane_parallel(path):
    """Every step of the reference's MT19937-driven episodes, one lane per recorded step:
    lane k is put in the state the reference was in before its step k and fed the uniform the
    reference drew there (and, where the reference called reset() first, the reset uniform)."""
    g = np.load(path)
    n = len(g["obs"])
    reset_mask = g["reset_before"].copy(); reset_mask[0] = 1
    u_r = np.where(g["reset_before"] == 1, g["u_reset"], 0.0); u_r[0] = g["u_first_reset"]
    t_before = np.zeros(n, np.uint8); c = 0
    for k in range(n):
        if reset_mask[k]:
            c = 0
        t_before[k] = c; c += 1
    prev = np.vstack([np.array([[1, 2, 2, 4, 0]], np.int8), g["state"][:-1]])
    keep = reset_mask == 0          # lanes the reference did not reset start from its previous state
    park = np.array([1, 2, 2, 4, 0], np.int8)
    init = np.where(keep[:, None], prev, park[None, :])
    b = SoccerBatch(n, 5, 4, float(g["slip"]))
    b.set_state(init[:, 0], init[:, 1], init[:, 2], init[:, 3], init[:, 4], t=t_before, needs_reset=0)
    m = b.alloc(n, np.uint8).upload(reset_mask); u = b.alloc(n, np.float64).upload(u_r)
    obs0 = b.alloc(n, np.uint16)
    b.reset(mask=m, u_reset=u, obs=obs0)
    got0 = obs0.download()
    assert got0[0] == g["first_obs"]
    rb = g["reset_before"] == 1
    np.testing.assert_array_equal(got0[rb], g["reset_obs"][rb])
    io = Bufs(b, with_u=True)
    out = io.step(g["actions"][:, 0], g["actions"][:, 1], u_step=g["u_step"], u_reset=np.zeros(n))
    np.testing.assert_array_equal(out["obs"], g["obs"])
    np.testing.assert_array_equal(out["reward"].astype(np.float64), g["reward_a"])
    np.testing.assert_array_equal(out["terminated"], g["terminated"])
    np.testing.assert_array_equal(out["truncated"], g["truncated"])
    np.testing.assert_array_equal(np.round(b.prob_table[out["prob_code"]], 2), g["p"])
    s = b.get_state()
    got = np.stack([s["row_a"], s["col_a"], s["row_b"], s["col_b"], s["poss"]], 1)
    np.testing.assert_array_equal(got, g["state"])
    b.close()


@pytest.mark.parametrize("slip,width,height,epw,n", [
    (0.0, 5, 4, 0, 65536),      # BASELINE config 2
    (0.2, 5, 4, 0, 65536),
    (0.0, 5, 4, 8, 65536 + 13), # ragged tail, 8 lanes per thread
    (0.2, 5, 4, 1, 4099),
    (0.3, 7, 5, 4, 8191),
    (1.0, 5, 4, 4, 8192),
    (0.0, 11, 7, 4, 8192),
    (0.2, 11, 7, 4, 4096),      # integer slip decision on a pitch whose tables do not fit the LDS
    (0.1, 6, 4, 4, 4100),       # slip with one near-integer threshold (slip_int = 2)
])
def test_config2_autoreset_every_lane_every_step_vs_oracle(slip, width, height, epw, n):
    """BASELINE config 2: 65 536 lanes, Philox seed 0, 200 steps (forces truncation at t=100 and
    auto-reset), actions from default_rng(2024); every lane, every step compared with the oracle."""
    steps = 200 if n >= 65536 else 120
    rng = np.random.default_rng(2024)
    acts = rng.integers(0, 5, size=(steps, 2, n), dtype=np.int8)
    seed, off = 0, 12345678901
    b = SoccerBatch(n, width, height, slip, seed=seed, autoreset=True, lane_offset=off, envs_per_thread=epw)
    o = Oracle(width, height, slip, n=n, seed=seed, lane_offset=off, autoreset=True)
    obs = b.alloc(n, np.uint16)
    b.reset(obs=obs)
    np.testing.assert_array_equal(obs.download(), o.reset())
    assert_state_equal(b, o)
    io = Bufs(b)
    last = np.zeros(n, np.int8)
    for k in range(steps):
        g = io.step(acts[k, 0], acts[k, 1])
        c = o.step(acts[k, 0], acts[k, 1])
        assert_out_equal(g, c)
        fin = (c["terminated"] | c["truncated"]) == 1
        last[fin] = c["reward"][fin]
        if k % 25 == 0 or k == steps - 1:
            assert_state_equal(b, o)
    np.testing.assert_array_equal(io.last.download(), last)
    hist, misuse = b.stats()
    np.testing.assert_array_equal(hist, o.hist)
    assert misuse == 0 and b.tick == o.tick == steps + 1
    assert hist.sum() > n          # at least one full episode per lane on average
    b.close()


@pytest.mark.parametrize("slip,width,height", [(0.2, 5, 4), (0.5, 5, 4), (1.0, 5, 4), (0.3, 7, 5), (0.25, 5, 4)])
def test_slip_thresholds_hit_exactly_and_within_rounding_distance(slip, width, height):
    """The slip kernel decides most draws from nominal thresholds and walks the exact float64 running
    sum only when u is within 2^-40 of one.  Probe both sides of that switch: uniforms exactly on,
    one ulp / 1e-15 / 1e-13 / 2e-12 / 1e-9 around every threshold of real transition lists."""
    o_tab = Oracle(width, height, slip, n=1)
    lut, kind, *_ = o_tab.tables()
    W = width + 2
    rng = np.random.default_rng(17)
    live = np.flatnonzero(kind == 1)
    states, acts, us = [], [], []
    deltas = [0.0, 1e-15, -1e-15, 1e-13, -1e-13, 2e-12, -2e-12, 1e-9, -1e-9]
    for f in rng.choice(live, 260, replace=False):
        p_ = f & 1; r = f >> 1; yb = r % W; r //= W; xb = r % height; r //= height; ya = r % W; xa = r // W
        aa, ab = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        probs, *_ = o_tab.transitions([xa, ya, xb, yb, p_], aa, ab)
        for thr in np.cumsum(probs):
            for d in deltas:
                for u in (thr + d, np.nextafter(thr, 0.0) if d == 0.0 else None, np.nextafter(thr, 2.0) if d == 0.0 else None):
                    if u is None or not (0.0 <= u < 1.0):
                        continue
                    states.append((xa, ya, xb, yb, p_)); acts.append((aa, ab)); us.append(u)
    st = np.array(states, np.int8); ac = np.array(acts, np.int8); u = np.array(us, np.float64)
    n = len(u)
    assert n > 5000
    b = SoccerBatch(n, width, height, slip)
    o = Oracle(width, height, slip, n=n)
    for x in (b, o):
        x.set_state(st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], t=0, needs_reset=np.zeros(n, np.uint8))
    got = b.step_host(ac[:, 0], ac[:, 1], u_step=u)
    want = o.step(ac[:, 0], ac[:, 1], u_step=u)
    for k in ("obs", "reward", "terminated", "truncated", "prob_code"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert_state_equal(b, o)
    b.close()


@pytest.mark.parametrize("width,height,slip,learner", [(5, 4, 0.0, None), (5, 4, 0.2, None), (7, 5, 0.3, None),
                                                       (5, 4, 0.2, "player_a"), (5, 4, 0.2, "player_b")])
def test_scalar_step_and_reset_match_oracle(width, height, slip, learner):
    """soccer_step_scalar / soccer_reset_scalar (inputs as kernel arguments, result polled from a mapped
    record): every field against the oracle's one-lane step from random reachable tuples (live and goal),
    with and without a fixed policy; the lane's resident state follows."""
    rng = np.random.default_rng(3)
    o = Oracle(width, height, slip, n=1)
    lut, kind, *_ = o.tables()
    W = width + 2
    b = SoccerBatch(1, width, height, slip)
    policy = None
    if learner:
        policy = rng.integers(0, 5, b.nS).astype(np.int8)
        b.set_policy("player_b" if learner == "player_a" else "player_a", policy)
    reach = np.flatnonzero(kind != 0)
    for f in rng.choice(reach, 400):
        p_ = f & 1; r = f >> 1; yb = r % W; r //= W; xb = r % height; r //= height; ya = r % W; xa = r // W
        st = (int(xa), int(ya), int(xb), int(yb), int(p_))
        aa, ab = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        t = int(rng.choice([0, 5, 98, 99]))
        u = float(rng.random())
        s_now = 0 if kind[f] == 2 else int(lut[f])
        if learner == "player_a": ab = int(policy[s_now])
        if learner == "player_b": aa = int(policy[s_now])
        o.set_state([st[0]], [st[1]], [st[2]], [st[3]], [st[4]], t=t, needs_reset=np.zeros(1, np.uint8))
        want = o.step(np.array([aa], np.int8), np.array([ab], np.int8), u_step=np.array([u]))
        got = b.step_scalar(st, None if learner == "player_b" else aa, None if learner == "player_a" else ab, u, t=t)
        for k in ("obs", "reward", "terminated", "truncated", "prob_code"):
            assert got[k] == int(want[k][0]), (k, st, aa, ab, u)
        assert got["state"] == (o.row_a[0], o.col_a[0], o.row_b[0], o.col_b[0], o.poss[0] & 1)
        assert got["t"] == o.t[0] and got["needs_reset"] == (o.poss[0] >> 1)
        assert_state_equal(b, o)
    for u in (0.0, 0.24, 0.25, 0.5, 0.74, 0.75, 0.999999):
        want = o.reset(u_reset=np.array([u]))
        got = b.reset_scalar(u)
        assert got["obs"] == int(want[0]) and got["t"] == 0 and got["needs_reset"] == 0
        assert_state_equal(b, o)
    # misuse and validation happen on the host, before any launch
    with pytest.raises(AssertionError, match="outside the pitch"):
        b.step_scalar((height, 1, 0, 2, 0), 0, 0, 0.5)
    with pytest.raises(AssertionError, match="unreachable"):
        b.step_scalar((0, 1, 0, 1, 0), 0, 0, 0.5)
    if not learner:
        with pytest.raises(AssertionError, match="actions"):
            b.step_scalar((0, 1, 0, 2, 0), 5, 0, 0.5)
    b2 = SoccerBatch(2, width, height, slip)
    with pytest.raises(AssertionError, match="n_lanes == 1"):
        b2.step_scalar((0, 1, 0, 2, 0), 0, 0, 0.5)
    b.close(); b2.close()


def test_host_mapped_handle_matches_a_device_resident_one():
    """SOCCER_F_HOST_MAPPED: state and staging live in pinned host memory the GPU accesses in place."""
    n = 1000
    rng = np.random.default_rng(8)
    bm = SoccerBatch(n, 5, 4, 0.2, seed=4, autoreset=True, host_mapped=True)
    bd = SoccerBatch(n, 5, 4, 0.2, seed=4, autoreset=True)
    np.testing.assert_array_equal(bm.reset_host(), bd.reset_host())
    for _ in range(120):
        a, c = rng.integers(0, 5, n, dtype=np.int8), rng.integers(0, 5, n, dtype=np.int8)
        g1, g2 = bm.step_host(a, c), bd.step_host(a, c)
        for k in ("obs", "reward", "terminated", "truncated", "prob_code", "final_obs"):
            np.testing.assert_array_equal(g1[k], g2[k], err_msg=k)
    sv = bm.host_state_view()
    s = bd.get_state()
    np.testing.assert_array_equal(sv[0, :n].view(np.int8), s["row_a"])
    np.testing.assert_array_equal(sv[5, :n], s["t"])
    bm.close(); bd.close()


def _outcomes_per_combination(o_tab, st, aa, ab, slip):
    """Number of collision outcomes (1, 2 or 4; 0 = inactive) of each of the 9 slip combinations of a
    transition list, recovered from the list's probabilities (weight * 1, .5 or .25 per entry)."""
    s = np.float64(slip); om = np.float64(1) - s
    w = [om * om, (om * s) * 0.5, (s * om) * 0.5, (s * s) * 0.25]
    cls = [0, 1, 1, 2, 2, 3, 3, 3, 3]
    probs = o_tab.transitions(st, aa, ab)[0]
    out, i = [], 0
    for c in range(9):
        wc = w[cls[c]]
        if wc == 0:
            out.append(0); continue
        n = 1 if probs[i] == wc else (2 if probs[i] == wc * 0.5 else 4)
        assert probs[i] == wc * (1.0, 0.5, None, 0.25)[n - 1]
        out.append(n); i += n
    assert i == len(probs)
    return out


def test_philox_draws_on_slip_thresholds_match_the_float64_cumsum():
    """The slip kernels decide "running sum <= u" on integers for Philox draws (u = (m + 1/2) * 2^-30: m >= c).  Replay
    draws whose m sits ON an integer threshold c or next to it (found by tools/find_threshold_draws.py in the handle's
    own Philox stream) from states where that threshold separates two outcomes; the oracle walks the float64
    running sum like categorical_sample.  Both the step and the rollout kernels.  Includes slip 0.1's draws on its
    mathematically dyadic threshold 27/32 and slips 0.15 / 0.4 / 2/3, whose list shapes round differently at such a
    threshold — all of them plain integer handles since ABI 3 (csrc/soccer_slip.hpp)."""
    import json
    d = json.load(open(os.path.join(GOLDEN, "threshold_draws.json")))
    seed = d["seed"]
    rng = np.random.default_rng(6)
    tabs, checked, kinds = {}, 0, set()
    for h in d["hits"]:
        slip, g, tick = h["slip"], h["lane"], h["tick"]
        if slip not in tabs:
            ot = Oracle(5, 4, slip, n=1); lut, kind, *_ = ot.tables()
            tabs[slip] = (ot, np.flatnonzero(kind == 1))
        ot, live = tabs[slip]
        lo, j = g & ~3, g & 3
        b = SoccerBatch(4, 5, 4, slip, seed=seed, lane_offset=lo)
        o = Oracle(5, 4, slip, n=4, seed=seed, lane_offset=lo)
        for name, c in h["thresholds"]:
            want_n = {"end": None, "two": 2}.get(name, 4)
            trials = attempts = 0
            while trials < 6:
                attempts += 1
                assert attempts < 20000, "no state found where combination %d has %s outcomes" % (c, want_n)
                fl = rng.choice(live, 4)
                p_ = fl & 1; r = fl >> 1; yb = r % 7; r //= 7; xb = r % 4; r //= 4; ya = r % 7; xa = r // 7
                a = rng.integers(0, 5, 4).astype(np.int8); bb = rng.integers(0, 5, 4).astype(np.int8)
                if want_n is not None:
                    n_c = _outcomes_per_combination(ot, (xa[j], ya[j], xb[j], yb[j], p_[j]), int(a[j]), int(bb[j]), slip)
                    if n_c[c] != want_n:
                        continue
                trials += 1
                for mode in ("step", "rollout"):
                    for x in (b, o):
                        x.set_state(xa, ya, xb, yb, p_, t=0, needs_reset=np.zeros(4, np.uint8))
                    b._check(b.lib.soccer_set_tick(b.h, tick)); o.tick = tick
                    # the lane's word really is the recorded draw
                    if mode == "step":
                        got = b.step_host(a, bb)
                    else:
                        A = b.alloc(4, np.int8).upload(a); B = b.alloc(4, np.int8).upload(bb)
                        ob = b.alloc(4, np.uint16); rw = b.alloc(4, np.int8); te = b.alloc(4, np.uint8); tr = b.alloc(4, np.uint8)
                        b.rollout(1, A, B, act_stride=4, obs=ob, reward=rw, terminated=te, truncated=tr, out_stride=4)
                        got = {"obs": ob.download(), "reward": rw.download(), "terminated": te.download(), "truncated": tr.download()}
                    want = o.step(a, bb)
                    for k in ("obs", "reward", "terminated", "truncated"):
                        np.testing.assert_array_equal(got[k], want[k], err_msg="%s %s %r" % (mode, k, h))
                    assert_state_equal(b, o)
                    checked += 1
            kinds.add(name)
        b.close()
    assert checked > 300 and kinds == {"end", "two", "four1", "four2", "four3"}
    assert any(h["slip"] == 0.1 and h["m"] == 27 * 2 ** 25 for h in d["hits"])      # slip 0.1: draws exactly on 27/32


@pytest.mark.parametrize("learner", ["player_a", "player_b"])
def test_single_agent_tables_every_row(learner):
    """Fixed-opponent mode against the REFERENCE's single-agent transition table (all ~43 000 rows):
    one lane per table row, put in the row's state, fed a uniform from the middle of the row's
    probability interval; the kernel (which looks the opponent's action up from the policy by the
    current observation) must land on exactly that row's next state / reward / done."""
    g = np.load(os.path.join(GOLDEN, "single_5x4_s0p2_%s.npz" % learner))
    rows, prob = g["rows"], g["prob"]      # xa,ya,xb,yb,p, a, k, nxa,nya,nxb,nyb,np, reward, done
    n = len(rows)
    starts = np.flatnonzero(rows[:, 6] == 0)
    cum = np.zeros(n); lo = np.zeros(n)
    for s_, e_ in zip(starts, np.append(starts[1:], n)):
        c = np.cumsum(prob[s_:e_]); cum[s_:e_] = c; lo[s_:e_] = np.concatenate([[0.0], c[:-1]])
    u = (lo + np.minimum(cum, 1.0)) / 2
    assert (u > lo).all() and (u < cum).all()
    b = SoccerBatch(n, 5, 4, 0.2)
    b.set_policy("player_b" if learner == "player_a" else "player_a", g["policy"])
    b.set_state(rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4], t=0, needs_reset=0)
    acts = rows[:, 5].astype(np.int8)
    out = b.step_host(acts if learner == "player_a" else None, acts if learner == "player_b" else None, u_step=u)
    s = b.get_state()
    got = np.stack([s["row_a"], s["col_a"], s["row_b"], s["col_b"], s["poss"]], 1)
    np.testing.assert_array_equal(got, rows[:, 7:12])
    sign = 1 if learner == "player_a" else -1          # the reference's learner-B table stores -r (:243-244)
    np.testing.assert_array_equal(sign * out["reward"], rows[:, 12])
    np.testing.assert_array_equal(out["terminated"], rows[:, 13])
    np.testing.assert_array_equal(b.prob_table[out["prob_code"]], prob)
    with pytest.raises(AssertionError, match="Both players"):
        b.set_policy(learner, g["policy"])
    b.close()


def test_single_agent_rollout_and_step_agree_with_oracle_policy_gather():
    n, T = 8192, 90
    rng = np.random.default_rng(12)
    policy = rng.integers(0, 5, size=761).astype(np.int8)
    acts = rng.integers(0, 5, size=(T, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, 0.2, seed=6, autoreset=True); b.set_policy("player_b", policy)
    b2 = SoccerBatch(n, 5, 4, 0.2, seed=6, autoreset=True); b2.set_policy("player_b", policy)
    o = Oracle(5, 4, 0.2, n=n, seed=6, autoreset=True)
    cur = o.reset(); b.reset(); b2.reset()
    A = b.alloc((T, n), np.int8).upload(acts); obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8)
    b.rollout(T, A, None, act_stride=n, obs=obs, reward=rew, out_stride=n)
    O, R = obs.download(), rew.download()
    for k in range(T):
        c = o.step(acts[k], policy[cur])                 # the oracle side does the gather on the host
        g2 = b2.step_host(acts[k], None)
        np.testing.assert_array_equal(O[k], c["obs"]); np.testing.assert_array_equal(R[k], c["reward"])
        np.testing.assert_array_equal(g2["obs"], c["obs"]); np.testing.assert_array_equal(g2["reward"], c["reward"])
        cur = c["obs"]
    assert_state_equal(b, o); assert_state_equal(b2, o)
    b.close(); b2.close()


# slips 0.2 / 0.3 / 0.05 take the integer threshold decision (KernelParams::CB), 0.5 too (dyadic: exact sums); 0.1 and 0.9
# have one scaled threshold within 2^-10 of an integer: integer decision except for a lane that draws exactly that integer
@pytest.mark.parametrize("slip,n,off", [(0.0, 65536, 0), (0.0, 65536 + 6, 1 << 33), (0.2, 32768, 4 * 123457), (0.0, 5, 0), (0.2, 3, 8),
                                        (0.3, 16384, 0), (0.05, 16384, 8), (0.5, 16384, 0), (0.1, 8192, 0), (0.9, 8192, 4)])
def test_hot_instantiation_plain_step_vs_oracle(slip, n, off):
    """The instantiation bench.py times: lane offset a multiple of 4 (one Philox block per thread), dword
    I/O, only the four mandatory outputs (LEAN), no step statistics — every lane, every step."""
    steps = 130
    rng = np.random.default_rng(77)
    b = SoccerBatch(n, 5, 4, slip, seed=9, autoreset=True, lane_offset=off, step_stats=False)
    o = Oracle(5, 4, slip, n=n, seed=9, lane_offset=off, autoreset=True)
    b.reset(); o.reset()
    aa = b.alloc(n, np.int8); ab = b.alloc(n, np.int8)
    obs = b.alloc(n, np.uint16); rew = b.alloc(n, np.int8); term = b.alloc(n, np.uint8); trunc = b.alloc(n, np.uint8)
    for k in range(steps):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        aa.upload(a[0]); ab.upload(a[1])
        b.step_plain(aa, ab, obs, rew, term, trunc)
        c = o.step(a[0], a[1])
        np.testing.assert_array_equal(obs.download(), c["obs"]); np.testing.assert_array_equal(rew.download(), c["reward"])
        np.testing.assert_array_equal(term.download(), c["terminated"]); np.testing.assert_array_equal(trunc.download(), c["truncated"])
    assert_state_equal(b, o)
    assert b.stats()[0].sum() == 0          # statistics are opt-in for single steps
    b.close()


@pytest.mark.parametrize("slip,fixed,n", [(0.0, "player_b", 16384), (0.2, "player_a", 16384 + 3), (0.1, "player_b", 8192), (0.5, "player_a", 4096)])
def test_hot_instantiation_with_fixed_policy_vs_oracle(slip, fixed, n):
    """Single-agent handles through the plain 8-argument batched_step (the fixed side's stream is NULL): the
    hot kernel with the policy lookup, every lane and step against the oracle, which is handed the fixed
    side's actions gathered on the host from the current observations."""
    steps = 120
    rng = np.random.default_rng(13)
    policy = rng.integers(0, 5, size=761).astype(np.int8)
    b = SoccerBatch(n, 5, 4, slip, seed=17, autoreset=True, step_stats=False)
    b.set_policy(fixed, policy)
    o = Oracle(5, 4, slip, n=n, seed=17, autoreset=True)
    obs = b.alloc(n, np.uint16); rew = b.alloc(n, np.int8); term = b.alloc(n, np.uint8); trunc = b.alloc(n, np.uint8)
    act = b.alloc(n, np.int8)
    b.reset(obs=obs); cur = o.reset()
    np.testing.assert_array_equal(obs.download(), cur)
    for k in range(steps):
        a = rng.integers(0, 5, n, dtype=np.int8)
        act.upload(a)
        if fixed == "player_b":
            b.step_plain(act, None, obs, rew, term, trunc); c = o.step(a, policy[cur])
        else:
            b.step_plain(None, act, obs, rew, term, trunc); c = o.step(policy[cur], a)
        np.testing.assert_array_equal(obs.download(), c["obs"], err_msg="obs step %d" % k)
        np.testing.assert_array_equal(rew.download(), c["reward"]); np.testing.assert_array_equal(term.download(), c["terminated"])
        np.testing.assert_array_equal(trunc.download(), c["truncated"])
        cur = c["obs"]
    assert_state_equal(b, o)
    b.close()


def test_no_autoreset_freezes_finished_lanes_and_flags_misuse():
    n = 4096
    rng = np.random.default_rng(7)
    b = SoccerBatch(n, 5, 4, 0.2, seed=3, autoreset=False)
    o = Oracle(5, 4, 0.2, n=n, seed=3, autoreset=False)
    # stepping before reset: every lane needs reset (reference: AssertionError at :376)
    io = Bufs(b)
    a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
    assert_out_equal(io.step(a[0], a[1]), o.step(a[0], a[1]))
    assert b.stats()[1] == 1 and o.misuse == n
    b.reset_stats(); o.misuse = 0
    obs = b.alloc(n, np.uint16); b.reset(obs=obs)
    np.testing.assert_array_equal(obs.download(), o.reset())
    for k in range(130):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        assert_out_equal(io.step(a[0], a[1]), o.step(a[0], a[1]))
    assert_state_equal(b, o)
    s = b.get_state()
    assert s["needs_reset"].all()          # everyone terminated or hit t=100 by now
    assert b.stats()[1] == 1
    # masked reset brings back only the selected lanes
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    m = b.alloc(n, np.uint8).upload(mask)
    b.reset(mask=m, obs=obs)
    np.testing.assert_array_equal(obs.download(), o.reset(mask=mask))
    assert_state_equal(b, o)
    b.close()


def test_rollout_equals_successive_steps_and_sampled_actions_match_oracle():
    n, T = 32768 + 5, 64
    rng = np.random.default_rng(11)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    for slip in (0.0, 0.2, 0.3, 0.5):
        b = SoccerBatch(n, 5, 4, slip, seed=99, autoreset=True)
        o = Oracle(5, 4, slip, n=n, seed=99, autoreset=True)
        b.reset(); o.reset()
        A = b.alloc((T, n), np.int8).upload(acts[:, 0]); B = b.alloc((T, n), np.int8).upload(acts[:, 1])
        obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8)
        term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
        rs = b.alloc(n, np.int32).fill(0); ec = b.alloc(n, np.int32).fill(0)
        b.rollout(T, A, B, act_stride=n, obs=obs, reward=rew, terminated=term, truncated=trunc,
                  out_stride=n, return_sum=rs, episode_count=ec)
        O, R, TE, TR = obs.download(), rew.download(), term.download(), trunc.download()
        ret = np.zeros(n, np.int64); eps = np.zeros(n, np.int64)
        for k in range(T):
            c = o.step(acts[k, 0], acts[k, 1])
            np.testing.assert_array_equal(O[k], c["obs"]); np.testing.assert_array_equal(R[k], c["reward"])
            np.testing.assert_array_equal(TE[k], c["terminated"]); np.testing.assert_array_equal(TR[k], c["truncated"])
            ret += c["reward"]; eps += (c["terminated"] | c["truncated"])
        assert_state_equal(b, o)
        np.testing.assert_array_equal(rs.download(), ret); np.testing.assert_array_equal(ec.download(), eps)
        np.testing.assert_array_equal(b.stats()[0], o.hist)
        # in-kernel uniform-random actions (BASELINE config 5 shape): same Philox stream in the oracle
        b.rollout(T, sample_actions=True, return_sum=rs, episode_count=ec)
        for k in range(T):
            a, bb = o.sample_actions()
            c = o.step(a, bb)
            ret += c["reward"]; eps += (c["terminated"] | c["truncated"])
        assert_state_equal(b, o)
        np.testing.assert_array_equal(rs.download(), ret); np.testing.assert_array_equal(ec.download(), eps)
        np.testing.assert_array_equal(b.stats()[0], o.hist)
        assert b.tick == o.tick
        b.close()


def _rollout_vs_oracle(b, o, acts_a, acts_b, policy=None, fixed=None):
    """T fused steps against T oracle steps: every output of every lane and step, state, histogram."""
    T, n = (acts_a if acts_a is not None else acts_b).shape
    A = None if acts_a is None else b.alloc((T, n), np.int8).upload(acts_a)
    B = None if acts_b is None else b.alloc((T, n), np.int8).upload(acts_b)
    obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8)
    term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    rs = b.alloc(n, np.int32).fill(0); ec = b.alloc(n, np.int32).fill(0)
    hist0 = int(o.hist.sum())
    cur = None
    if fixed:                                   # the oracle side gathers the fixed side's action on the host
        lut = o.tables()[0]
        f = ((((o.row_a.astype(np.int64) * o.W + o.col_a) * o.H + o.row_b) * o.W + o.col_b) << 1) | (o.poss & 1)
        cur = lut[f]
    b.rollout(T, A, B, act_stride=n, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n,
              return_sum=rs, episode_count=ec)
    O, R, TE, TR = obs.download(), rew.download(), term.download(), trunc.download()
    ret = np.zeros(n, np.int64)
    for k in range(T):
        a = policy[cur] if fixed == "player_a" else acts_a[k]
        bb = policy[cur] if fixed == "player_b" else acts_b[k]
        c = o.step(a, bb)
        np.testing.assert_array_equal(O[k], c["obs"], err_msg="obs step %d" % k)
        np.testing.assert_array_equal(R[k], c["reward"], err_msg="reward step %d" % k)
        np.testing.assert_array_equal(TE[k], c["terminated"], err_msg="terminated step %d" % k)
        np.testing.assert_array_equal(TR[k], c["truncated"], err_msg="truncated step %d" % k)
        ret += c["reward"]
        cur = c["obs"]
    assert_state_equal(b, o)
    np.testing.assert_array_equal(rs.download(), ret)
    np.testing.assert_array_equal(b.stats()[0], o.hist)
    assert int(ec.download().sum()) == int(o.hist.sum()) - hist0


@pytest.mark.parametrize("width,height,slip", [(5, 4, 0.0), (6, 4, 0.0), (7, 5, 0.0), (5, 4, 0.2), (6, 4, 0.3), (5, 4, 0.5)])
def test_table_rollout_special_lanes_and_no_autoreset(width, height, slip):
    """batched_rollout with lanes frozen on entry, lanes injected into goal tuples, mixed with ordinary lanes inside one
    thread's group of four; handles without auto-reset; a ragged lane count (which sends the whole call to the per-lane
    rollout kernel — the byte-parallel one gets the same treatment in tests/test_gpu_swar.py).  (Named after round 1's
    LDS transition-table kernel, which the byte-parallel rollout replaced.)"""
    rng = np.random.default_rng(21)
    n, T = 4099, 130
    ot = Oracle(width, height, slip, n=1)
    lut, kind, *_ = ot.tables()
    W = width + 2
    def tuples(fl):
        p_ = fl & 1; r = fl >> 1; yb = r % W; r //= W; xb = r % height; r //= height; ya = r % W; xa = r // W
        return xa, ya, xb, yb, p_
    live, goal = np.flatnonzero(kind == 1), np.flatnonzero(kind == 2)
    for autoreset in (True, False):
        b = SoccerBatch(n, width, height, slip, seed=31, autoreset=autoreset)
        o = Oracle(width, height, slip, n=n, seed=31, autoreset=autoreset)
        # a third of the lanes in goal tuples (needs_reset 0: absorbing step), a third frozen, the rest live
        fl = np.where(rng.random(n) < 0.33, rng.choice(goal, n), rng.choice(live, n))
        nr = (rng.random(n) < 0.33).astype(np.uint8)
        t0 = rng.choice([0, 3, 97, 99, 100], n).astype(np.uint8)
        xa, ya, xb, yb, p_ = tuples(fl)
        for x in (b, o):
            x.set_state(xa, ya, xb, yb, p_, t=t0, needs_reset=nr)
        acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
        _rollout_vs_oracle(b, o, acts[:, 0], acts[:, 1])
        assert b.stats()[1] == 1                         # frozen lanes were stepped: sticky misuse flag
        b.reset_stats()
        # and from a clean reset (auto-reset on: the steady-state loop; off: lanes freeze as they finish)
        b.reset(); o.reset(); o.hist[:] = 0; b.reset_stats()
        _rollout_vs_oracle(b, o, acts[:, 0], acts[:, 1])
        assert b.stats()[1] == (0 if autoreset else 1)
        b.close()


@pytest.mark.parametrize("fixed", ["player_a", "player_b"])
def test_table_rollout_single_agent_policy(fixed):
    n, T = 2048, 110
    rng = np.random.default_rng(4)
    policy = rng.integers(0, 5, size=761).astype(np.int8)
    acts = rng.integers(0, 5, size=(T, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, 0.0, seed=8, autoreset=True); b.set_policy(fixed, policy)
    o = Oracle(5, 4, 0.0, n=n, seed=8, autoreset=True)
    b.reset(); o.reset()
    _rollout_vs_oracle(b, o, None if fixed == "player_a" else acts, None if fixed == "player_b" else acts,
                       policy=policy, fixed=fixed)
    b.close()


def test_config5_selfplay_rollout_histogram_matches_cpu_exactly():
    """BASELINE config 5: 2^20 lanes x 100-step horizon, both players sampling from mixed policy tables
    in-kernel (per-lane Philox), reward histogram compared with the CPU oracle — exact counts, since
    every lane is bit-exact.  (The reference has no Minimax-Q; the policies here are random mixed
    strategies of the shape a Minimax-Q learner would hold: [nS, 5] per player.)"""
    n, T = 1 << 20, 100
    rng = np.random.default_rng(94)
    probs_a = rng.dirichlet(np.ones(5) * 0.7, size=761); probs_b = rng.dirichlet(np.ones(5) * 0.7, size=761)
    ta = SoccerBatch.mixed_policy_thresholds(probs_a); tb = SoccerBatch.mixed_policy_thresholds(probs_b)
    b = SoccerBatch(n, 5, 4, 0.0, seed=1994, autoreset=True)
    o = Oracle(5, 4, 0.0, n=n, seed=1994, autoreset=True)
    obs0 = b.alloc(n, np.uint16); b.reset(obs=obs0)
    cur = o.reset()
    np.testing.assert_array_equal(obs0.download(), cur)
    da = b.alloc(ta.shape, np.uint16).upload(ta); db = b.alloc(tb.shape, np.uint16).upload(tb)
    rs = b.alloc(n, np.int32).fill(0); ec = b.alloc(n, np.int32).fill(0)
    b.reset_stats()
    b.rollout(T, sample_actions=True, mix_a=da, mix_b=db, return_sum=rs, episode_count=ec)
    ret = np.zeros(n, np.int64); eps = np.zeros(n, np.int64)
    for k in range(T):
        a, bb = o.sample_actions_mixed(cur, ta, tb)
        c = o.step(a, bb)
        ret += c["reward"]; eps += (c["terminated"] | c["truncated"]); cur = c["obs"]
    hist = b.stats()[0]
    np.testing.assert_array_equal(hist, o.hist)
    np.testing.assert_array_equal(rs.download(), ret); np.testing.assert_array_equal(ec.download(), eps)
    assert_state_equal(b, o)
    assert hist.sum() > 2 * n and hist[0] > 0 and hist[2] > 0
    b.close()


def test_results_do_not_depend_on_sharding_or_vector_width():
    """Lanes [0,N) on one handle == two handles of N/2 with lane_offset (multi-GPU contract)."""
    n, T = 16384, 40
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    def run(lo, hi, epw):
        b = SoccerBatch(hi - lo, 5, 4, 0.2, seed=42, autoreset=True, lane_offset=lo, envs_per_thread=epw)
        b.reset(); io = Bufs(b); outs = []
        for k in range(T):
            outs.append(io.step(acts[k, 0, lo:hi], acts[k, 1, lo:hi])["obs"])
        h = b.stats()[0]; b.close()
        return np.stack(outs), h
    full, hf = run(0, n, 4)
    a, ha = run(0, n // 2, 8); c, hc = run(n // 2, n, 1)
    np.testing.assert_array_equal(full, np.concatenate([a, c], 1))
    np.testing.assert_array_equal(hf, ha + hc)


def test_config4_full_size_eight_shards_equal_one_handle_and_invariants():
    """BASELINE config 4 at full size on one GPU: 8 388 608 lanes as ONE handle vs 8 shards of 2^20 with
    global lane offsets (what 8 GPUs would each own) — identical per-lane results — plus the
    size-independent invariants of the outputs."""
    n, G, T = 1 << 23, 8, 30
    per = n // G
    big = SoccerBatch(n, 5, 4, 0.0, seed=2024, autoreset=True)
    big.reset()
    rs = big.alloc(n, np.int32).fill(0); ec = big.alloc(n, np.int32).fill(0)
    obs = big.alloc((T, n), np.uint16); rew = big.alloc((T, n), np.int8)
    term = big.alloc((T, n), np.uint8); trunc = big.alloc((T, n), np.uint8)
    big.rollout(T, sample_actions=True, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n,
                return_sum=rs, episode_count=ec)
    R_big, E_big, S_big = rs.download(), ec.download(), big.get_state()
    O, RW, TE, TR = obs.download(), rew.download(), term.download(), trunc.download()
    hist_big = big.stats()[0]
    del obs, rew, term, trunc
    big.close()
    hist_sum = np.zeros(3, np.uint64)
    for g in range(G):
        b = SoccerBatch(per, 5, 4, 0.0, seed=2024, autoreset=True, lane_offset=g * per)
        b.reset()
        r = b.alloc(per, np.int32).fill(0); e = b.alloc(per, np.int32).fill(0)
        b.rollout(T, sample_actions=True, return_sum=r, episode_count=e)
        sl = slice(g * per, (g + 1) * per)
        np.testing.assert_array_equal(r.download(), R_big[sl]); np.testing.assert_array_equal(e.download(), E_big[sl])
        s = b.get_state()
        for k in ("row_a", "col_a", "row_b", "col_b", "poss", "t"):
            np.testing.assert_array_equal(s[k], S_big[k][sl], err_msg=k)
        hist_sum += b.stats()[0]
        b.close()
    np.testing.assert_array_equal(hist_sum, hist_big)
    # invariants of the step outputs at full size
    fin = (TE | TR).astype(bool)
    assert int(fin.sum()) == int(hist_big.sum()) == int(E_big.sum())
    assert ((RW != 0) <= (TE == 1)).all()                      # a reward only on the scoring step
    assert int((RW == 1).sum()) == int(hist_big[2]) and int((RW == -1).sum()) == int(hist_big[0])
    assert not TR.any()                                        # 30 steps < 100: nothing truncates
    assert np.isin(O[fin], (253, 254, 435, 436)).all()         # auto-reset: a fresh episode's first observation
    assert (O > 0).all() and (O < 761).all()
    assert int(R_big.sum()) == int(RW.astype(np.int64).sum())
    np.testing.assert_array_equal(S_big["t"] <= T, True)


def test_config3_full_size_step_path_equals_rollout_path_and_oracle_on_a_shard(monkeypatch):
    """BASELINE config 3 at full size: 2^20 lanes x 1000 steps of uniform-random joint actions.  The single-step
    kernel (byte-parallel rules, what bench.py times) and the fused rollout — here forced onto the per-lane kernel
    that walks the rule tables (SOCCER_ROLLOUT=1) — are two independent implementations: every output of every lane
    and step must agree, and the first 4 096 lanes are checked against the CPU oracle step by step (results do not
    depend on the sharding)."""
    import torch
    n, K, sub = 1 << 20, 1000, 4096
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    acts = torch.randint(0, 5, (K, 2, n), dtype=torch.int8, device=dev, generator=g)
    def bufs():
        return (torch.empty((K, n), dtype=torch.int16, device=dev), torch.empty((K, n), dtype=torch.int8, device=dev),
                torch.empty((K, n), dtype=torch.uint8, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev))
    b1 = SoccerBatch(n, 5, 4, 0.0, seed=33, autoreset=True, step_stats=False)
    b1.reset()
    o1 = bufs()
    for k in range(K):
        b1.step_plain(acts[k, 0], acts[k, 1], o1[0][k], o1[1][k], o1[2][k], o1[3][k])
    b1.sync()
    monkeypatch.setenv("SOCCER_ROLLOUT", "1")             # read at soccer_create: this handle rolls out through the rule tables
    b2 = SoccerBatch(n, 5, 4, 0.0, seed=33, autoreset=True)
    monkeypatch.delenv("SOCCER_ROLLOUT")
    b2.reset()
    o2 = bufs()
    a_flat = acts.view(K * 2, n)
    b2.rollout(K, a_flat[0], a_flat[1], act_stride=2 * n, obs=o2[0], reward=o2[1], terminated=o2[2], truncated=o2[3], out_stride=n)
    b2.sync()
    for x, y, name in zip(o1, o2, ("obs", "reward", "terminated", "truncated")):
        assert torch.equal(x, y), name
    s1, s2 = b1.get_state(), b2.get_state()
    for kx in s1:
        np.testing.assert_array_equal(s1[kx], s2[kx], err_msg=kx)
    assert b1.tick == b2.tick == K + 1
    # size-independent properties of the outputs
    fin = (o1[2] | o1[3]) != 0
    assert int(o1[0].max()) < b1.nS and int(o1[0].min()) >= 1            # auto-reset: never the terminal index
    assert bool(((o1[1] != 0) <= (o1[2] != 0)).all())                     # a reward only on a terminating step
    hist = b2.stats()[0]
    assert int(hist.sum()) == int(fin.sum()) and int(hist[2]) == int((o1[1] == 1).sum()) and int(hist[0]) == int((o1[1] == -1).sum())
    # the oracle on the first shard
    o = Oracle(5, 4, 0.0, n=sub, seed=33, autoreset=True)
    o.reset()
    A = acts[:, :, :sub].cpu().numpy()
    O = o1[0][:, :sub].cpu().numpy().view(np.uint16); R = o1[1][:, :sub].cpu().numpy()
    TE = o1[2][:, :sub].cpu().numpy(); TR = o1[3][:, :sub].cpu().numpy()
    for k in range(K):
        c = o.step(A[k, 0], A[k, 1])
        np.testing.assert_array_equal(O[k], c["obs"]); np.testing.assert_array_equal(R[k], c["reward"])
        np.testing.assert_array_equal(TE[k], c["terminated"]); np.testing.assert_array_equal(TR[k], c["truncated"])
    b1.close(); b2.close()


def test_graph_capture_replays_advance_the_tick():
    n, T = 8192, 6
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, 0.0, seed=1, autoreset=True)
    o = Oracle(5, 4, 0.0, n=n, seed=1, autoreset=True)
    b.reset(); o.reset()
    A = b.alloc((T, n), np.int8).upload(acts[:, 0]); B = b.alloc((T, n), np.int8).upload(acts[:, 1])
    obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8)
    term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    b.graph_begin()
    for k in range(T):
        b.step_plain(A.row(k), B.row(k), obs.row(k), rew.row(k), term.row(k), trunc.row(k))
    g = b.graph_end()
    for rep in range(3):
        b.graph_launch(g, 1)
        O, R = obs.download(), rew.download()
        for k in range(T):
            c = o.step(acts[k, 0], acts[k, 1])
            np.testing.assert_array_equal(O[k], c["obs"]); np.testing.assert_array_equal(R[k], c["reward"])
    assert b.tick == o.tick == 1 + 3 * T
    assert_state_equal(b, o)
    b.graph_destroy(g)
    # odd numbers of captured calls (1 and 5; round 3 refused them), interleaved with eager steps and replayed twice in a row
    k_next = [0]

    def expect(rows):
        O, R = obs.download(), rew.download()
        for k in rows:
            c = o.step(acts[k, 0], acts[k, 1])
            np.testing.assert_array_equal(O[k], c["obs"]); np.testing.assert_array_equal(R[k], c["reward"])
    b.graph_begin(); b.step_plain(A.row(0), B.row(0), obs.row(0), rew.row(0), term.row(0), trunc.row(0)); g1 = b.graph_end()
    b.graph_begin()
    for k in range(1, 6):
        b.step_plain(A.row(k), B.row(k), obs.row(k), rew.row(k), term.row(k), trunc.row(k))
    g5 = b.graph_end()
    for rep in range(2):
        b.graph_launch(g1, 1); expect([0])
        b.step_plain(A.row(3), B.row(3), obs.row(3), rew.row(3), term.row(3), trunc.row(3)); expect([3])     # an eager step in between
        b.graph_launch(g5, 1); expect(range(1, 6))
        b.graph_launch(g5, 2); b.sync()
        for _ in range(2):
            for k in range(1, 6):
                c = o.step(acts[k, 0], acts[k, 1])
        np.testing.assert_array_equal(obs.download()[5], c["obs"])
        b.graph_launch(g1, 1); b.graph_launch(g1, 1); o.step(acts[0, 0], acts[0, 1]); expect([0])
    assert b.tick == o.tick
    assert_state_equal(b, o)
    b.graph_destroy(g1); b.graph_destroy(g5)
    b.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_graph_capture_of_rollouts_and_policy_steps(slip):
    """A captured sequence mixing two fused rollouts and two single-agent steps (policy lookup in the kernel) replays
    exactly like the eager sequence on the oracle; ticks advance per replay."""
    n, T = 4096, 9
    rng = np.random.default_rng(8)
    policy = rng.integers(0, 5, size=761).astype(np.int8)
    acts = rng.integers(0, 5, size=(2 * T + 2, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, slip, seed=2, autoreset=True, step_stats=False); b.set_policy("player_b", policy)
    o = Oracle(5, 4, slip, n=n, seed=2, autoreset=True)
    b.reset(); cur = o.reset()
    A = b.alloc(acts.shape, np.int8).upload(acts)
    obs = b.alloc((2 * T + 2, n), np.uint16); rew = b.alloc((2 * T + 2, n), np.int8)
    term = b.alloc((2 * T + 2, n), np.uint8); trunc = b.alloc((2 * T + 2, n), np.uint8)
    b.graph_begin()
    b.rollout(T, A.row(0), None, act_stride=n, obs=obs.row(0), reward=rew.row(0), terminated=term.row(0), truncated=trunc.row(0), out_stride=n)
    b.step_plain(A.row(T), None, obs.row(T), rew.row(T), term.row(T), trunc.row(T))
    b.rollout(T, A.row(T + 1), None, act_stride=n, obs=obs.row(T + 1), reward=rew.row(T + 1), terminated=term.row(T + 1),
              truncated=trunc.row(T + 1), out_stride=n)
    b.step_plain(A.row(2 * T + 1), None, obs.row(2 * T + 1), rew.row(2 * T + 1), term.row(2 * T + 1), trunc.row(2 * T + 1))
    g = b.graph_end()
    for rep in range(3):
        b.graph_launch(g, 1)
        O, R, TE = obs.download(), rew.download(), term.download()
        for k in range(2 * T + 2):
            c = o.step(acts[k], policy[cur])
            np.testing.assert_array_equal(O[k], c["obs"], err_msg="rep %d step %d" % (rep, k))
            np.testing.assert_array_equal(R[k], c["reward"]); np.testing.assert_array_equal(TE[k], c["terminated"])
            cur = c["obs"]
    assert b.tick == o.tick == 1 + 3 * (2 * T + 2)
    assert_state_equal(b, o)
    b.graph_destroy(g); b.close()


def test_state_injection_rejects_unreachable_tuples():
    b = SoccerBatch(2, 5, 4, 0.0)
    with pytest.raises(KeyError):
        b.set_state([1, 1], [1, 1], [1, 2], [1, 2], [0, 0], t=0, needs_reset=0)   # both players in one cell
    with pytest.raises(KeyError):
        b.set_state([0, 1], [0, 1], [1, 2], [3, 3], [0, 0], t=0, needs_reset=0)   # corner of a goal column
    b.close()
