"""GPU tests of the byte-parallel kernels (csrc/soccer_swar.hpp: step_kernel_swar, rollout_swar_kernel) through the C ABI:
every instantiation (lean / full outputs, slip 0 / integer slip selection, fixed policy) against the oracle, every lane,
every step, on the reference's pitch sizes; the fallbacks they sit next to; and what happens to action bytes outside
0..4 on every device-pointer path (reference: IndexError at soccer_simultaneous_env.py:393)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gym_soccer_littman94_amd import SoccerBatch, VectorSoccerEnv
from oracle.oracle import Oracle


def _state_equal(b, o):
    s = b.get_state()
    for k, v in (("row_a", o.row_a), ("col_a", o.col_a), ("row_b", o.row_b), ("col_b", o.col_b), ("poss", o.poss & 1),
                 ("needs_reset", (o.poss >> 1) & 1), ("t", o.t)):
        np.testing.assert_array_equal(s[k], v, err_msg=k)


class _IO:
    def __init__(self, b, full):
        n = b.n
        self.b, self.full = b, full
        self.aa = b.alloc(n, np.int8); self.ab = b.alloc(n, np.int8)
        self.obs = b.alloc(n, np.uint16); self.rew = b.alloc(n, np.int8)
        self.term = b.alloc(n, np.uint8); self.trunc = b.alloc(n, np.uint8)
        self.code = b.alloc(n, np.uint8) if full else None
        self.fin = b.alloc(n, np.uint16) if full else None
        # the gym surface's outputs (ABI 2): float32 rewards of both players and terminated | truncated, from the same launch
        self.rfa = b.alloc(n, np.float32).fill(7) if full else None
        self.rfb = b.alloc(n, np.float32).fill(7) if full else None
        self.done = b.alloc(n, np.uint8).fill(7) if full else None

    def step(self, a, bb):
        if a is not None: self.aa.upload(a)
        if bb is not None: self.ab.upload(bb)
        self.b.step(self.aa if a is not None else None, self.ab if bb is not None else None, obs=self.obs, reward=self.rew,
                    terminated=self.term, truncated=self.trunc, prob_code=self.code, final_obs=self.fin,
                    reward_a_f32=self.rfa, reward_b_f32=self.rfb, finished=self.done)
        out = dict(obs=self.obs.download(), reward=self.rew.download(), terminated=self.term.download(), truncated=self.trunc.download())
        if self.full:
            out.update(prob_code=self.code.download(), final_obs=self.fin.download())
            _check_gym_outputs(out, self.rfa.download(), self.rfb.download(), self.done.download())
        return out


def _check_gym_outputs(out, rfa, rfb, done):
    """reward_a_f32 / reward_b_f32 / finished are the int8 reward and the two flags in the types a gym caller reads
    (:400-404): exactly +-1.0 / 0.0, B = 0 - A with positive zeros, finished = terminated | truncated"""
    r = out["reward"].astype(np.float32)
    np.testing.assert_array_equal(rfa.view(np.uint32), r.view(np.uint32))
    np.testing.assert_array_equal(rfb.view(np.uint32), (np.float32(0) - r).view(np.uint32))
    np.testing.assert_array_equal(done, out["terminated"] | out["truncated"])


def _check(got, exp, k):
    for key in got:
        np.testing.assert_array_equal(got[key], exp[key], err_msg="%s at step %d" % (key, k))


@pytest.mark.parametrize("w,h,slip,n", [(5, 4, 0.0, 16384 + 4), (5, 4, 0.2, 16384), (6, 4, 0.5, 8192), (7, 5, 0.3, 8192 + 8),
                                        (9, 6, 0.0, 8192), (11, 7, 0.0, 8192), (11, 7, 0.2, 8192), (5, 4, 1.0, 4096), (5, 4, 0.05, 4096),
                                        (5, 4, 0.1, 8192), (7, 5, 0.9, 4096)])
@pytest.mark.parametrize("autoreset", [True, False])
@pytest.mark.parametrize("full", [True, False])
def test_step_kernel_swar_every_lane_every_step(w, h, slip, n, autoreset, full):
    """lean (4 outputs) and full (+ prob_code, final_obs, episode histogram) instantiations; without auto-reset the lanes
    freeze one after the other, so the frozen-lane code runs on a growing share of them"""
    steps = 140
    rng = np.random.default_rng(int(slip * 100) + w + 2 * full)
    b = SoccerBatch(n, w, h, slip, seed=17, autoreset=autoreset, lane_offset=4 * 1000003, step_stats=full)
    o = Oracle(w, h, slip, n=n, seed=17, autoreset=autoreset, lane_offset=4 * 1000003)
    io = _IO(b, full)
    a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
    _check(io.step(a[0], a[1]), o.step(a[0], a[1]), -1)               # before any reset: every lane frozen
    assert b.stats()[1] == SoccerBatch.MISUSE_FROZEN
    b.reset_stats(); o.hist[:] = 0
    b.reset(); o.reset()
    for k in range(steps):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        _check(io.step(a[0], a[1]), o.step(a[0], a[1]), k)
    _state_equal(b, o)
    hist, misuse = b.stats()
    if full:
        np.testing.assert_array_equal(hist, o.hist)
        assert hist.sum() > 0
    else:
        assert hist.sum() == 0
    assert misuse == (0 if autoreset else SoccerBatch.MISUSE_FROZEN) and b.tick == o.tick
    b.close()


@pytest.mark.parametrize("w,h,slip,fixed", [(5, 4, 0.0, "player_b"), (5, 4, 0.2, "player_a"), (7, 5, 0.3, "player_b"), (11, 7, 0.0, "player_a"),
                                            (5, 4, 0.1, "player_a")])
@pytest.mark.parametrize("full", [True, False])
def test_step_kernel_swar_fixed_policy(w, h, slip, fixed, full):
    n, steps = 8192, 110
    rng = np.random.default_rng(3)
    o = Oracle(w, h, slip, n=n, seed=23, autoreset=True)
    policy = rng.integers(0, 5, size=o.nS).astype(np.int8)
    b = SoccerBatch(n, w, h, slip, seed=23, autoreset=True, step_stats=False)
    b.set_policy(fixed, policy)
    io = _IO(b, full)
    obs0 = b.alloc(n, np.uint16); b.reset(obs=obs0)
    cur = o.reset()
    np.testing.assert_array_equal(obs0.download(), cur)
    for k in range(steps):
        act = rng.integers(0, 5, size=n, dtype=np.int8)
        a, bb = (policy[cur], act) if fixed == "player_a" else (act, policy[cur])
        c = o.step(a, bb)
        _check(io.step(None if fixed == "player_a" else act, None if fixed == "player_b" else act), c, k)
        cur = c["obs"]
    _state_equal(b, o)
    b.close()


_EXTRAS = [0]


def _rollout_vs_oracle(b, o, acts, T, n, sample=False, mix=None, extras=None):
    """extras: also ask for batched_rollout_ex's per-step final_obs / prob_code and check them; None = every other call of this
    helper, so that both the plain and the FULL instantiations see every parametrisation family"""
    if extras is None:
        _EXTRAS[0] += 1; extras = bool(_EXTRAS[0] & 1)
    A = B = None
    if not sample:
        A = b.alloc((T, n), np.int8).upload(acts[:, 0]); B = b.alloc((T, n), np.int8).upload(acts[:, 1])
    obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8); term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    rs = b.alloc(n, np.int32).fill(0); ec = b.alloc(n, np.int32).fill(0)
    lut = o.tables()[0]
    f = ((((o.row_a.astype(np.int64) * o.W + o.col_a) * o.H + o.row_b) * o.W + o.col_b) << 1) | (o.poss & 1)
    cur = lut[f]
    da = db = None
    if mix is not None:
        da = b.alloc(mix[0].shape, np.uint16).upload(mix[0]); db = b.alloc(mix[1].shape, np.uint16).upload(mix[1])
    fo = b.alloc((T, n), np.uint16).fill(0xEE) if extras else None; cd = b.alloc((T, n), np.uint8).fill(0xEE) if extras else None
    b.rollout(T, A, B, act_stride=n, sample_actions=sample, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n,
              return_sum=rs, episode_count=ec, mix_a=da, mix_b=db, final_obs=fo, prob_code=cd)
    O, R, TE, TR = obs.download(), rew.download(), term.download(), trunc.download()
    FO = fo.download() if extras else None; CD = cd.download() if extras else None
    ret = np.zeros(n, np.int64); eps = np.zeros(n, np.int64)
    for k in range(T):
        if sample:
            a, bb = o.sample_actions_mixed(cur, *(mix if mix is not None else (None, None)))
        else:
            a, bb = acts[k, 0], acts[k, 1]
        c = o.step(a, bb)
        np.testing.assert_array_equal(O[k], c["obs"], err_msg="obs %d" % k); np.testing.assert_array_equal(R[k], c["reward"], err_msg="reward %d" % k)
        np.testing.assert_array_equal(TE[k], c["terminated"]); np.testing.assert_array_equal(TR[k], c["truncated"])
        if extras:
            np.testing.assert_array_equal(FO[k], c["final_obs"], err_msg="final_obs %d" % k)
            np.testing.assert_array_equal(CD[k], c["prob_code"], err_msg="prob_code %d" % k)
        frozen = ((o.poss >> 1) & 1).astype(bool) if not o.autoreset else np.zeros(n, bool)
        ret += c["reward"]; eps += ((c["terminated"] | c["truncated"]) != 0)
        cur = c["obs"]
    _state_equal(b, o)
    np.testing.assert_array_equal(rs.download(), ret)
    np.testing.assert_array_equal(b.stats()[0], o.hist)
    return ec.download(), eps


@pytest.mark.parametrize("w,h,slip", [(5, 4, 0.0), (7, 5, 0.0), (9, 6, 0.0), (11, 7, 0.0), (5, 4, 0.2), (11, 7, 0.3), (6, 4, 1.0), (5, 4, 0.1),
                                      (5, 4, 0.02), (5, 4, 0.999)])      # the last two: thresholds crowd a bucket, one-by-one selection
def test_rollout_swar_streams_every_lane_every_step(w, h, slip):
    n, T = 8192, 120
    rng = np.random.default_rng(w + int(10 * slip))
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    for autoreset in (True, False):
        b = SoccerBatch(n, w, h, slip, seed=4, autoreset=autoreset); o = Oracle(w, h, slip, n=n, seed=4, autoreset=autoreset)
        b.reset(); o.reset()
        ec, eps = _rollout_vs_oracle(b, o, acts, T, n)
        if autoreset:
            np.testing.assert_array_equal(ec, eps)
            assert eps.sum() > n // 2
        else:
            assert b.stats()[1] == SoccerBatch.MISUSE_FROZEN          # lanes that finished were stepped again: left untouched
        # a second rollout continues from the state the first one left (frozen lanes included), goal tuples injected
        lut, kind, gv, isd, isdp = o.tables()
        goal = np.flatnonzero(kind == 2)[:64]
        st = b.get_state()
        for name, div in (("poss", 1), ("col_b", 2), ("row_b", 2 * o.W), ("col_a", 2 * o.W * o.H), ("row_a", 2 * o.W * o.H * o.W)):
            mod = {"poss": 2, "col_b": o.W, "row_b": o.H, "col_a": o.W, "row_a": 1 << 30}[name]
            st[name][:64] = (goal // div) % mod
        st["needs_reset"][:64] = 0; st["t"][:64] = 5
        b.set_state(st["row_a"], st["col_a"], st["row_b"], st["col_b"], st["poss"], t=st["t"], needs_reset=st["needs_reset"])
        o.set_state(st["row_a"], st["col_a"], st["row_b"], st["col_b"], st["poss"], t=st["t"], needs_reset=st["needs_reset"])
        b.reset_stats(); o.hist[:] = 0
        _rollout_vs_oracle(b, o, acts[:40], 40, n)
        b.close()


@pytest.mark.parametrize("w,h,slip", [(5, 4, 0.0), (5, 4, 0.2), (11, 7, 0.0), (5, 4, 0.9)])
def test_rollout_swar_sampled_and_mixed_policies(w, h, slip):
    """in-kernel sampling: uniform, and from [nS, 4] mixed-policy thresholds — staged in LDS on 5x4, gathered from global
    memory on 11x7 (nS = 11 705 rows do not fit next to each other in 64 KB)"""
    n, T = 8192, 100
    rng = np.random.default_rng(31)
    o = Oracle(w, h, slip, n=n, seed=8, autoreset=True)
    mix = (SoccerBatch.mixed_policy_thresholds(rng.dirichlet(np.ones(5) * 0.6, size=o.nS)),
           SoccerBatch.mixed_policy_thresholds(rng.dirichlet(np.ones(5) * 0.6, size=o.nS)))
    for m in (None, mix):
        b = SoccerBatch(n, w, h, slip, seed=8, autoreset=True); o = Oracle(w, h, slip, n=n, seed=8, autoreset=True)
        b.reset(); o.reset()
        ec, eps = _rollout_vs_oracle(b, o, None, T, n, sample=True, mix=m)
        np.testing.assert_array_equal(ec, eps)
        b.close()


def test_fallback_rollout_still_matches(monkeypatch):
    """SOCCER_ROLLOUT=1 keeps the per-lane rollout kernel (what slips like 0.1 and pitches beyond the byte arithmetic take)"""
    monkeypatch.setenv("SOCCER_ROLLOUT", "1")
    n, T = 4096, 60
    rng = np.random.default_rng(2)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    for slip in (0.0, 0.2, 0.1):
        b = SoccerBatch(n, 5, 4, slip, seed=4, autoreset=True); o = Oracle(5, 4, slip, n=n, seed=4, autoreset=True)
        b.reset(); o.reset()
        _rollout_vs_oracle(b, o, acts, T, n)
        b.close()


# ---- action bytes outside 0..4 ---------------------------------------------------------------------------------------------
BAD = np.array([5, 6, 7, 8, 127, -1, -128, -123], np.int8)


def _reachable(b, o):
    lut, kind, *_ = o.tables()
    s = b.get_state()
    ra, ca, rb, cb, p = (s[k].astype(np.int64) for k in ("row_a", "col_a", "row_b", "col_b", "poss"))
    assert (ra >= 0).all() and (ra < o.H).all() and (rb >= 0).all() and (rb < o.H).all()
    assert (ca >= 0).all() and (ca < o.W).all() and (cb >= 0).all() and (cb < o.W).all()
    assert (kind[(((ra * o.W + ca) * o.H + rb) * o.W + cb) * 2 + p] != 0).all()


@pytest.mark.parametrize("path", ["swar_lean", "swar_full", "generic", "swar_slip_0p1", "hot_max_steps_200", "rollout_swar", "rollout_fallback"])
def test_bad_action_bytes_on_device_paths_stay_inside_the_tables_and_are_flagged(path, monkeypatch):
    n, T = 8192, 30
    slip = 0.1 if path == "swar_slip_0p1" else 0.0
    ms = 200 if path == "hot_max_steps_200" else 100          # max_steps > 127 does not fit the byte arithmetic: per-lane hot kernel
    if path == "rollout_fallback":
        monkeypatch.setenv("SOCCER_ROLLOUT", "1")
    rng = np.random.default_rng(6)
    b = SoccerBatch(n, 5, 4, slip, seed=1, autoreset=True, step_stats=False, max_steps=ms)
    o = Oracle(5, 4, slip, n=n, seed=1, autoreset=True, max_steps=ms)
    b.reset()
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    poison = rng.random((T, 2, n)) < 0.02
    acts[poison] = BAD[rng.integers(0, len(BAD), size=int(poison.sum()))]
    A = b.alloc((T, n), np.int8).upload(acts[:, 0]); B = b.alloc((T, n), np.int8).upload(acts[:, 1])
    obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8); term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    code = b.alloc(n, np.uint8); last = b.alloc(n, np.int8).fill(0)
    assert b.stats()[1] == 0
    if path.startswith("rollout"):
        b.rollout(T, A, B, act_stride=n, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n)
    else:
        for k in range(T):
            kw = {}
            if path == "swar_full": kw = dict(prob_code=code)
            if path == "generic": kw = dict(prob_code=code, last_return=last)
            b.step(A.row(k), B.row(k), obs=obs.row(k), reward=rew.row(k), terminated=term.row(k), truncated=trunc.row(k), **kw)
    assert b.stats()[1] == SoccerBatch.MISUSE_ACTION and b.peek_misuse() == SoccerBatch.MISUSE_ACTION
    _reachable(b, o)
    O = obs.download()
    assert (O < b.nS).all() and (np.abs(rew.download()) <= 1).all()
    # the semantics: a byte executes as (byte & 7) with 5..7 -> NOOP — i.e. exactly what the oracle does on the mapped action
    canon = acts.view(np.uint8) & 7
    canon[canon > 4] = 0
    o.reset()
    for k in range(T):
        c = o.step(canon[k, 0].astype(np.int8), canon[k, 1].astype(np.int8))
        np.testing.assert_array_equal(O[k], c["obs"], err_msg="step %d" % k)
    _state_equal(b, o)
    b.reset_stats()
    assert b.stats()[1] == 0
    b.close()


@pytest.mark.parametrize("w,h,slip,ms", [(13, 9, 0.0, 100), (13, 9, 0.2, 100), (5, 4, 0.0, 200), (5, 4, 0.2, 250)])
def test_handles_beyond_the_byte_arithmetic_take_the_per_lane_kernels(w, h, slip, ms):
    """H * W > 128 (13x9: 9 x 15 = 135 cells) or max_steps > 127: step_kernel_hot / rollout_kernel through the rule tables"""
    n, steps, T = 8192, 60, 60
    rng = np.random.default_rng(w + ms)
    b = SoccerBatch(n, w, h, slip, seed=5, autoreset=True, max_steps=ms, step_stats=False)
    o = Oracle(w, h, slip, n=n, seed=5, autoreset=True, max_steps=ms)
    io = _IO(b, False)
    b.reset(); o.reset()
    for k in range(steps):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        _check(io.step(a[0], a[1]), o.step(a[0], a[1]), k)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    b.reset_stats(); o.hist[:] = 0                   # the single steps above did not feed the histogram (step_stats off)
    _rollout_vs_oracle(b, o, acts, T, n)
    # single-agent mode on such a handle: the fixed side's action by the generic kernel's policy gather
    policy = rng.integers(0, 5, size=o.nS).astype(np.int8)
    b.set_policy("player_b", policy)
    lut = o.tables()[0]
    f = ((((o.row_a.astype(np.int64) * o.W + o.col_a) * o.H + o.row_b) * o.W + o.col_b) << 1) | (o.poss & 1)
    cur = lut[f]
    for k in range(30):
        act = rng.integers(0, 5, size=n, dtype=np.int8)
        c = o.step(act, policy[cur])
        _check(io.step(act, None), c, k)
        cur = c["obs"]
    _state_equal(b, o)
    b.close()


@pytest.mark.parametrize("w,h,slip,n,shift", [(13, 9, 0.0, 4096, 0), (13, 9, 0.2, 4096, 0), (5, 4, 0.0, 4099, 0), (5, 4, 0.2, 4098, 0),
                                              (5, 4, 0.0, 4096, 1), (5, 4, 0.2, 4101, 3), (5, 4, 0.33, 4096, 0)])
def test_gym_outputs_on_the_per_lane_and_unaligned_paths(w, h, slip, n, shift):
    """reward_a_f32 / reward_b_f32 / finished from the kernels next to the byte-parallel one: pitches beyond the byte
    arithmetic, slip values without an exact integer decision (0.33), ragged lane counts (a byte-I/O tail launch) and float
    streams that are only 4-byte aligned (the whole step falls back to byte I/O)"""
    rng = np.random.default_rng(n + shift)
    b = SoccerBatch(n, w, h, slip, seed=9, autoreset=True, step_stats=True)
    o = Oracle(w, h, slip, n=n, seed=9, autoreset=True)
    aa = b.alloc(n, np.int8); ab = b.alloc(n, np.int8)
    rew = b.alloc(n, np.int8); term = b.alloc(n, np.uint8); trunc = b.alloc(n, np.uint8); obs = b.alloc(n, np.uint16)
    rfa = b.alloc(n + 4, np.float32).fill(7); rfb = b.alloc(n + 4, np.float32).fill(7); done = b.alloc(n + 4, np.uint8).fill(7)
    b.reset(); o.reset()
    for k in range(130):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        aa.upload(a[0]); ab.upload(a[1])
        b.step(aa, ab, obs=obs, reward=rew, terminated=term, truncated=trunc,
               reward_a_f32=rfa.ptr + 4 * shift, reward_b_f32=rfb.ptr + 4 * shift, finished=done.ptr + shift)
        exp = o.step(a[0], a[1])
        out = dict(obs=obs.download(), reward=rew.download(), terminated=term.download(), truncated=trunc.download())
        _check(out, {key: exp[key] for key in out}, k)
        _check_gym_outputs(out, rfa.download()[shift:shift + n], rfb.download()[shift:shift + n], done.download()[shift:shift + n])
    # nothing was written outside [shift, shift + n)
    for arr, fill in ((rfa.download().view(np.uint32), 0x07070707), (rfb.download().view(np.uint32), 0x07070707), (done.download(), 7)):
        assert (arr[:shift] == fill).all() and (arr[shift + n:] == fill).all()
    _state_equal(b, o)
    np.testing.assert_array_equal(b.stats()[0], o.hist)
    b.close()


def test_gym_outputs_are_device_only():
    n = 64
    b = SoccerBatch(n, 5, 4, 0.0, seed=1)
    b.reset()
    a = np.zeros(n, np.int8); f = np.zeros(n, np.float32)
    from gym_soccer_littman94_amd._lib import StepArgs
    args = StepArgs(a.ctypes.data, a.ctypes.data, None, None, None, None, None, None, None, None, None, f.ctypes.data, None, None)
    import ctypes
    assert b.lib.batched_step_host(b.h, ctypes.byref(args)) != 0
    assert "device-only" in b.lib.soccer_last_error(b.h).decode()
    b.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
@pytest.mark.parametrize("fixed", [None, "player_a", "player_b"])
def test_vector_env_float_rewards_by_the_kernel_equal_the_lazy_casts(slip, fixed):
    """VectorSoccerEnv(io="device"): float_rewards=True (the step kernel writes the float32 rewards and
    infos["_final_observation"]) returns what float_rewards=False computes on access, for both agents and in single-agent mode"""
    import torch
    n = 4096
    rng = np.random.default_rng(5)
    kw = dict(slip_prob=slip, seed=3, io="device")
    nS = VectorSoccerEnv(4, slip_prob=slip).nS
    if fixed:
        kw[fixed + "_policy"] = rng.integers(0, 5, size=nS).astype(np.int8)
    v1 = VectorSoccerEnv(n, float_rewards=True, **kw); v2 = VectorSoccerEnv(n, float_rewards=False, **kw)
    o1 = v1.reset(); o2 = v2.reset()
    ags = v1.return_agent
    assert ags == v2.return_agent
    for k in range(120):
        act = {ag: torch.from_numpy(rng.integers(0, 5, size=n, dtype=np.int8)).cuda() for ag in ags}
        ob1, r1, te1, tr1, i1 = v1.step(act)
        ob2, r2, te2, tr2, i2 = v2.step(act)
        for ag in ags:
            assert r1[ag].dtype == torch.float32 and r1[ag].shape == (n,)
            assert torch.equal(r1[ag], r2[ag]) and torch.equal(ob1[ag], ob2[ag])
            assert torch.equal(r1[ag], (v1.reward_int8.float() if ag == "player_a" else 0.0 - v1.reward_int8.float()))
            assert torch.equal(te1[ag], te2[ag]) and torch.equal(tr1[ag], tr2[ag])
            assert torch.equal(i1[ag]["p"], i2[ag]["p"])
        assert i1["_final_observation"].dtype == torch.bool
        assert torch.equal(i1["_final_observation"], i2["_final_observation"])
        assert torch.equal(i1["_final_observation"], te1[ags[0]] | tr1[ags[0]])
    v1.close(); v2.close()


def test_bad_actions_on_host_paths_are_refused_before_any_launch():
    n = 1000
    b = SoccerBatch(n, 5, 4, 0.0, seed=1, autoreset=True)
    b.reset()
    tick = b.tick
    good = np.zeros(n, np.int8)
    for v in (5, 127, -1, -128):
        bad = good.copy(); bad[n - 3] = v
        with pytest.raises(AssertionError, match="actions must be in 0..4"):
            b.step_host(good, bad)
        with pytest.raises(AssertionError, match="player_a in lane %d" % (n - 3)):
            b.step_host(bad, good)
    assert b.tick == tick and b.stats()[1] == 0          # nothing was launched
    b.step_host(good, good)
    b.close()


def test_vector_env_strict_mode_reports_bad_device_actions():
    import torch
    n = 4096
    v = VectorSoccerEnv(n, seed=0, io="device")
    v.reset()
    ok = torch.zeros(n, dtype=torch.int8, device="cuda")
    bad = ok.clone(); bad[17] = -1
    v.step({"player_a": ok, "player_b": ok})
    v.step({"player_a": bad, "player_b": ok})
    torch.cuda.synchronize()
    with pytest.raises(AssertionError, match="actions must be in 0..4"):
        for _ in range(4):                                   # the host-mapped flags are looked at every fourth step
            v.step({"player_a": ok, "player_b": ok})
    for _ in range(8):
        v.step({"player_a": ok, "player_b": ok})             # the flag was cleared with the report
    v.close()
    vn = VectorSoccerEnv(64, seed=0)
    vn.reset()
    with pytest.raises(AssertionError, match="0..4"):
        vn.step({"player_a": np.full(64, 5), "player_b": np.zeros(64, np.int64)})
    vn.close()


def test_one_handle_above_2_22_lanes_counts_every_episode():
    """A single 2^23-lane handle stepped through batched_step_ex with the episode histogram on (VectorSoccerEnv's default):
    the byte-parallel step launches one wave per 256 lanes — 32 768 of them here — and every wave must own its histogram
    slot (round 2 had 16 384 slots and wave % 16384: two waves of one launch shared a slot without atomics).  The
    histogram must equal the count of terminated | truncated, split by the sign of the reward, over all outputs."""
    import torch
    n, steps = 1 << 23, 130
    dev = torch.device("cuda", 0)
    b = SoccerBatch(n, 5, 4, 0.0, seed=11, autoreset=True, step_stats=True)
    g = torch.Generator(device=dev); g.manual_seed(5)
    acts = torch.randint(0, 5, (16, 2, n), dtype=torch.int8, device=dev, generator=g)
    u16 = getattr(torch, "uint16", torch.int16)
    obs = torch.empty(n, dtype=u16, device=dev); rew = torch.empty(n, dtype=torch.int8, device=dev)
    term = torch.empty(n, dtype=torch.uint8, device=dev); trunc = torch.empty(n, dtype=torch.uint8, device=dev)
    fin = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    b.reset(); b.reset_stats()
    want = np.zeros(3, np.int64); ends = 0
    for k in range(steps):
        b.step(acts[k % 16, 0], acts[k % 16, 1], obs=obs, reward=rew, terminated=term, truncated=trunc, finished=fin)
        b.sync()
        done = (term | trunc) != 0
        assert bool((fin.bool() == done).all())
        for i, v in enumerate((-1, 0, 1)):
            want[i] += int(((rew == v) & done).sum())
        ends += int(done.sum())
        assert int(((rew != 0) & ~done).sum()) == 0          # a reward only on the step that ends an episode
    hist, misuse = b.stats()
    assert misuse == 0 and ends == want.sum() and want[1] > n // 2      # every lane truncated at least once in 130 steps
    np.testing.assert_array_equal(hist.astype(np.int64), want)
    b.close()


def test_last_return_rides_on_the_byte_parallel_step():
    """last_return (A's return of the lane's most recently finished episode) no longer forces the per-lane kernel: the
    byte-parallel step writes it with a read-modify-write of the thread's own dword, only on steps that end an episode;
    lanes whose episode goes on keep what the stream held.  Against the oracle, with a sentinel in the stream."""
    from oracle.oracle import Oracle
    n = 4096
    rng = np.random.default_rng(8)
    for slip in (0.0, 0.2):
        b = SoccerBatch(n, 5, 4, slip, seed=21, autoreset=True, step_stats=False)
        o = Oracle(5, 4, slip, n=n, seed=21, autoreset=True)
        A = b.alloc(n, np.int8); B = b.alloc(n, np.int8); rew = b.alloc(n, np.int8)
        last = b.alloc(n, np.int8).fill(0x55)
        b.reset(); o.reset()
        want = np.full(n, 0x55, np.int8)
        for _ in range(140):
            a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
            A.upload(a[0]); B.upload(a[1])
            b.step(A, B, reward=rew, last_return=last)
            c = o.step(a[0], a[1])
            done = (c["terminated"] | c["truncated"]) != 0
            want = np.where(done, c["reward"], want)
            np.testing.assert_array_equal(rew.download(), c["reward"])
            np.testing.assert_array_equal(last.download(), want)
        assert (want != 0x55).all()
        b.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_lean_device_vector_env_returns_what_the_full_one_does(slip):
    """VectorSoccerEnv(io="device", info=False): the step kernel's instantiation without final_obs / prob_code / histogram
    without the int8 reward stream and with one float32 reward stream (23 B per env-step; player_b's reward is the negation of
    player_a's, computed on access).  Same seed, same actions: observations, float32 rewards of
    both agents, terminated, truncated and infos["_final_observation"] equal the full env's at every step."""
    import torch
    n = 8192 + 4
    dev = torch.device("cuda", 0)
    full = VectorSoccerEnv(n, slip_prob=slip, seed=3, io="device")
    lean = VectorSoccerEnv(n, slip_prob=slip, seed=3, io="device", info=False)
    assert lean.reward_int8 is None
    of, _ = full.reset(); ol, _ = lean.reset()
    assert bool((of["player_a"] == ol["player_a"]).all())
    g = torch.Generator(device=dev); g.manual_seed(9)
    for k in range(150):
        a = torch.randint(0, 5, (2, n), dtype=torch.int8, device=dev, generator=g)
        act = {"player_a": a[0], "player_b": a[1]}
        o1, r1, te1, tr1, i1 = full.step(act)
        o2, r2, te2, tr2, i2 = lean.step(act)
        for ag in ("player_a", "player_b"):
            assert bool((o1[ag] == o2[ag]).all()) and bool((r1[ag] == r2[ag]).all())
            assert r2[ag].dtype == torch.float32
            assert bool((te1[ag] == te2[ag]).all()) and bool((tr1[ag] == tr2[ag]).all())
        assert bool((i1["_final_observation"] == i2["_final_observation"]).all())
        assert "final_observation" not in i2 and "player_a" not in i2
    assert lean.batch.misuse() == 0
    with pytest.raises(AssertionError):
        lean.episode_histogram()
    full.close(); lean.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_rollout_at_every_block_alignment_and_length(slip):
    """At slip 0 eight ticks share a Philox block, and the rollout fetches its action streams a block (8 ticks) ahead: both
    are steered by tick & 7.  Every alignment of the first tick (0..8) x lengths around the block size, the trajectories and
    the state against the oracle — and against T single steps from the same state (rollout step j = the j-th batched_step)."""
    n = 1024
    rng = np.random.default_rng(12)
    acts = rng.integers(0, 5, size=(20, 2, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, slip, seed=77, autoreset=True); o = Oracle(5, 4, slip, n=n, seed=77, autoreset=True)
    b2 = SoccerBatch(n, 5, 4, slip, seed=77, autoreset=True)
    b.reset(); o.reset(); b2.reset()
    A2 = b2.alloc(n, np.int8); B2 = b2.alloc(n, np.int8); O2 = b2.alloc(n, np.uint16)
    for tick0 in range(0, 9):
        for T in (1, 2, 7, 8, 9, 15, 16, 17, 20):
            t0 = 1000 * 8 + tick0
            b._check(b.lib.soccer_set_tick(b.h, t0)); b2._check(b2.lib.soccer_set_tick(b2.h, t0)); o.tick = t0
            st = b.get_state()
            b2.set_state(st["row_a"], st["col_a"], st["row_b"], st["col_b"], st["poss"], t=st["t"], needs_reset=st["needs_reset"])
            _rollout_vs_oracle(b, o, acts[:T], T, n)
            for k in range(T):
                A2.upload(acts[k, 0]); B2.upload(acts[k, 1])
                b2.step(A2, B2, obs=O2)
            s1, s2 = b.get_state(), b2.get_state()
            for key in s1:
                np.testing.assert_array_equal(s1[key], s2[key], err_msg="%s after tick0 %d T %d" % (key, tick0, T))
            assert b.tick == t0 + T == b2.tick
            b.reset_stats(); o.hist[:] = 0
    b.close(); b2.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_rollout_with_a_lane_count_that_is_not_a_multiple_of_four(slip):
    """n = 4 099 with row strides padded to 4 100: the byte-parallel rollout over the first 4 096 lanes, the last three through
    the per-lane kernel on the same ticks — trajectories, state, per-lane sums and the histogram against the oracle."""
    n, T, stride = 4099, 50, 4100
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    b = SoccerBatch(n, 5, 4, slip, seed=9, autoreset=True); o = Oracle(5, 4, slip, n=n, seed=9, autoreset=True)
    b.reset(); o.reset()
    pad = np.zeros((T, stride), np.int8)
    pa = pad.copy(); pa[:, :n] = acts[:, 0]; pb = pad.copy(); pb[:, :n] = acts[:, 1]
    A = b.alloc((T, stride), np.int8).upload(pa); B = b.alloc((T, stride), np.int8).upload(pb)
    obs = b.alloc((T, stride), np.uint16).fill(0); rew = b.alloc((T, stride), np.int8).fill(0)
    rs = b.alloc(n + 1, np.int32).fill(0)
    b.rollout(T, A, B, act_stride=stride, obs=obs, reward=rew, out_stride=stride, return_sum=rs)
    O, R = obs.download(), rew.download()
    ret = np.zeros(n, np.int64)
    for k in range(T):
        c = o.step(acts[k, 0], acts[k, 1])
        np.testing.assert_array_equal(O[k, :n], c["obs"], err_msg="obs %d" % k)
        np.testing.assert_array_equal(R[k, :n], c["reward"], err_msg="reward %d" % k)
        ret += c["reward"]
    assert (O[:, n:] == 0).all() and (R[:, n:] == 0).all()           # nothing written beyond the batch
    _state_equal(b, o)
    np.testing.assert_array_equal(rs.download()[:n], ret)
    np.testing.assert_array_equal(b.stats()[0], o.hist)
    assert b.tick == o.tick
    b.close()


def test_clock_stamps_and_the_captured_timer():
    """soccer_stamp: device clock stamps in a host-mapped block, eagerly and as graph nodes; a captured timer_start / _mark
    pair is stamps 0 / 1 and soccer_timer_read watches the closing one change.  Stamps grow along the stream, the captured
    timer agrees with them, and an eager timer (HIP events) around the same launches is of the same size."""
    n, K = 1 << 16, 8
    b = SoccerBatch(n, 5, 4, 0.0, seed=1, autoreset=True, step_stats=False)
    A = b.alloc(n, np.int8).fill(1); B = b.alloc(n, np.int8).fill(2)
    obs = b.alloc(n, np.uint16); rew = b.alloc(n, np.int8); te = b.alloc(n, np.uint8); tr = b.alloc(n, np.uint8)
    b.reset()
    b.stamp(10)
    for k in range(K):
        b.step_plain(A, B, obs, rew, te, tr); b.stamp(11 + k)
    b.sync()
    t, khz = b.stamps(10, K + 1)
    assert khz > 1000 and (np.diff(t.astype(np.int64)) > 0).all()
    b.graph_begin(); b.timer_start()
    for k in range(K):
        b.step_plain(A, B, obs, rew, te, tr)
    b.timer_mark(); g = b.graph_end()
    for _ in range(3):
        b.graph_launch(g, 1); ms = b.timer_read()
        s01, _ = b.stamps(0, 2)
        assert ms > 0 and abs(ms - (int(s01[1]) - int(s01[0])) / khz) < 1e-6
    b.sync()
    b.timer_start()
    for k in range(K):
        b.step_plain(A, B, obs, rew, te, tr)
    eager_ms = b.timer_stop()
    assert 0.2 * ms < eager_ms < 20 * ms
    with pytest.raises(AssertionError):
        b.stamp(256)
    assert b.tick == 1 + K * 5          # the reset, K eager steps, three replays of K, K eager steps (the capture itself consumes none)
    # a graph captured AFTER a stamped one has no timer nodes of its own: it is not "stamped", and the eager timer around its
    # replay is the HIP-event one (the sticky handle flag used to mark it, and timer_read then polled a slot nobody wrote)
    b.graph_begin()
    for k in range(K):
        b.step_plain(A, B, obs, rew, te, tr)
    g2 = b.graph_end()
    b.timer_start(); b.graph_launch(g2, 1); ms2 = b.timer_stop()
    assert 0.2 * ms < ms2 < 20 * ms
    # two replays of the stamped graph with no synchronisation in between: timer_read waits for the stream instead of trusting
    # a closing stamp sampled while the first replay was still in flight, and returns the LAST replay's duration
    b.graph_launch(g, 1); b.graph_launch(g, 1); ms3 = b.timer_read()
    s01, _ = b.stamps(0, 2)
    assert 0 < ms3 < 20 * ms and abs(ms3 - (int(s01[1]) - int(s01[0])) / khz) < 1e-6
    # a capture that marks without starting is refused even though an EARLIER capture had a timer
    b.graph_begin()
    with pytest.raises(RuntimeError, match="needs soccer_timer_start"):
        b.timer_mark()
    b.step_plain(A, B, obs, rew, te, tr); b.step_plain(A, B, obs, rew, te, tr)
    g3 = b.graph_end()
    b.graph_destroy(g); b.graph_destroy(g2); b.graph_destroy(g3); b.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_a_step_split_into_several_launches_is_the_same_step(slip, monkeypatch):
    """step_kernel_swar addresses every stream with 32-bit byte offsets, so a handle beyond 2^30 lanes is stepped by several
    launches on one tick, each handed its part of every stream (launch_step in soccer_hip.hip).  SOCCER_SWAR_LAUNCH_LANES
    (read by soccer_create) shrinks the part so that the split can be tested at a size the oracle finishes: 3 parts + a
    short one + a ragged 3-lane tail, every output stream, the histogram and last_return, eagerly and as a captured graph."""
    monkeypatch.setenv("SOCCER_SWAR_LAUNCH_LANES", "4096")
    n, T = 3 * 4096 + 1028 + 3, 6
    rng = np.random.default_rng(12)
    b = SoccerBatch(n, 5, 4, slip, seed=23, autoreset=True, lane_offset=4 * 77, step_stats=True)
    monkeypatch.delenv("SOCCER_SWAR_LAUNCH_LANES")
    o = Oracle(5, 4, slip, n=n, seed=23, autoreset=True, lane_offset=4 * 77)
    io = _IO(b, True)
    last = b.alloc(n, np.int8).fill(0x55); want = np.full(n, 0x55, np.int8)
    b.reset(); o.reset()
    for k in range(60):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        io.aa.upload(a[0]); io.ab.upload(a[1])
        b.step(io.aa, io.ab, obs=io.obs, reward=io.rew, terminated=io.term, truncated=io.trunc, prob_code=io.code, final_obs=io.fin,
               reward_a_f32=io.rfa, reward_b_f32=io.rfb, finished=io.done, last_return=last)
        c = o.step(a[0], a[1])
        got = dict(obs=io.obs.download(), reward=io.rew.download(), terminated=io.term.download(), truncated=io.trunc.download(),
                   prob_code=io.code.download(), final_obs=io.fin.download())
        _check(got, c, k)
        _check_gym_outputs(got, io.rfa.download(), io.rfb.download(), io.done.download())
        want = np.where((c["terminated"] | c["truncated"]) != 0, c["reward"], want)
        np.testing.assert_array_equal(last.download(), want)
    # captured: T steps per replay, each of them split the same way; the tick slot is published once per step
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    A2 = b.alloc((T, n + 5), np.int8); B2 = b.alloc((T, n + 5), np.int8)   # rows n + 5 apart: every row 8-byte aligned
    obs = b.alloc((T, n + 5), np.uint16); rew = b.alloc((T, n + 5), np.int8)
    term = b.alloc((T, n + 5), np.uint8); trunc = b.alloc((T, n + 5), np.uint8)
    pad = np.zeros((T, 5), np.int8)
    A2.upload(np.concatenate([acts[:, 0], pad], axis=1)); B2.upload(np.concatenate([acts[:, 1], pad], axis=1))
    b.graph_begin()
    for k in range(T):
        b.step_plain(A2.row(k), B2.row(k), obs.row(k), rew.row(k), term.row(k), trunc.row(k))
    g = b.graph_end()
    for rep in range(2):
        b.graph_launch(g, 1)
        O, R, TE, TR = obs.download(), rew.download(), term.download(), trunc.download()
        for k in range(T):
            c = o.step(acts[k, 0], acts[k, 1])
            np.testing.assert_array_equal(O[k, :n], c["obs"]); np.testing.assert_array_equal(R[k, :n], c["reward"])
            np.testing.assert_array_equal(TE[k, :n], c["terminated"]); np.testing.assert_array_equal(TR[k, :n], c["truncated"])
    b.graph_destroy(g)
    _state_equal(b, o)
    hist, misuse = b.stats()
    assert misuse == 0 and b.tick == o.tick
    b.close()


@pytest.mark.parametrize("slip", [0.1, 0.2, 0.9, 1.0, 2.0 / 3.0])
def test_both_slip_selections_of_the_single_step_agree(slip, monkeypatch):
    """step_kernel_swar<.., SLIPM, ..>: 1 compares the thresholds one by one, 2 reads the 10-bit bucket table each wave parks
    in LDS + two exact compares (slips within about [0.09, 0.96], and 1.0).  SOCCER_STEP_SLIP_ONE_BY_ONE (read by soccer_create)
    keeps a handle on the first form: same seed, same actions -> the same streams as the table form and as the oracle, on a lane
    count that leaves the last wave partly filled (its idle lanes still carry their piece of the table)."""
    n = 4096 + 4 * 37
    rng = np.random.default_rng(int(slip * 1000))
    monkeypatch.setenv("SOCCER_STEP_SLIP_ONE_BY_ONE", "1")
    b1 = SoccerBatch(n, 5, 4, slip, seed=5, autoreset=True, step_stats=True)
    monkeypatch.delenv("SOCCER_STEP_SLIP_ONE_BY_ONE")
    b2 = SoccerBatch(n, 5, 4, slip, seed=5, autoreset=True, step_stats=True)
    o = Oracle(5, 4, slip, n=n, seed=5, autoreset=True)
    io1, io2 = _IO(b1, True), _IO(b2, True)
    b1.reset(); b2.reset(); o.reset()
    for k in range(60):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        g1, g2, c = io1.step(a[0], a[1]), io2.step(a[0], a[1]), o.step(a[0], a[1])
        _check(g1, c, k); _check(g2, c, k)
    _state_equal(b1, o); _state_equal(b2, o)
    np.testing.assert_array_equal(b1.stats()[0], b2.stats()[0])
    b1.close(); b2.close()


@pytest.mark.parametrize("slip,fixed", [(0.0, None), (0.2, None), (0.0, "player_b"), (0.2, "player_a")])
def test_action_streams_with_the_non_temporal_hint(slip, fixed):
    """SOCCER_F_STREAM_ACTIONS only changes how step_kernel_swar reads its action streams (non-temporal instead of plain loads:
    two arms of the kernel): same seed, same actions -> the same streams as a default handle and as the oracle."""
    n = 8192 + 4 * 11
    rng = np.random.default_rng(41)
    b1 = SoccerBatch(n, 5, 4, slip, seed=6, autoreset=True, step_stats=True)
    b2 = SoccerBatch(n, 5, 4, slip, seed=6, autoreset=True, step_stats=True, stream_actions=True)
    o = Oracle(5, 4, slip, n=n, seed=6, autoreset=True)
    policy = rng.integers(0, 5, o.nS).astype(np.int8)
    if fixed:
        b1.set_policy(fixed, policy); b2.set_policy(fixed, policy)
    io1, io2 = _IO(b1, True), _IO(b2, True)
    b1.reset(); b2.reset(); cur = o.reset()
    for k in range(50):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        if fixed == "player_a": a[0] = policy[cur]
        if fixed == "player_b": a[1] = policy[cur]
        c = o.step(a[0], a[1])
        for io in (io1, io2):
            _check(io.step(None if fixed == "player_a" else a[0], None if fixed == "player_b" else a[1]), c, k)
        cur = c["obs"]
    _state_equal(b1, o); _state_equal(b2, o)
    b1.close(); b2.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
def test_a_rollout_split_into_several_launches_is_the_same_rollout(slip, monkeypatch):
    """rollout_swar_kernel addresses its streams with 32-bit byte offsets too: a handle beyond 2^30 lanes is rolled out part by
    part (batched_rollout in soccer_hip.hip), every part over the same ticks, the last one publishing the tick.
    SOCCER_SWAR_LAUNCH_LANES shrinks the part: 3 parts + a short one, streams in and four trajectories out, per-lane return sums
    / episode counts, then sampled actions; the histogram; followed by single steps on the same ticks."""
    monkeypatch.setenv("SOCCER_SWAR_LAUNCH_LANES", "4096")
    n, T = 3 * 4096 + 1028 + 3, 37           # + 3: a ragged tail behind SEVERAL parts (the per-lane kernel, absolute lane indices)
    b = SoccerBatch(n, 5, 4, slip, seed=29, autoreset=True, lane_offset=4 * 55, step_stats=True)
    monkeypatch.delenv("SOCCER_SWAR_LAUNCH_LANES")
    o = Oracle(5, 4, slip, n=n, seed=29, autoreset=True, lane_offset=4 * 55)
    rng = np.random.default_rng(13)
    b.reset(); o.reset()
    acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
    _rollout_vs_oracle(b, o, acts, T, n)
    _rollout_vs_oracle(b, o, None, T, n, sample=True)
    io = _IO(b, True)
    for k in range(5):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        _check(io.step(a[0], a[1]), o.step(a[0], a[1]), k)
    _state_equal(b, o)
    hist, misuse = b.stats()
    np.testing.assert_array_equal(hist, o.hist)
    assert misuse == 0 and b.tick == o.tick
    b.close()


@pytest.mark.parametrize("slip", [0.5, 0.2, 0.25])
@pytest.mark.parametrize("full", [False, True])
def test_caller_uniforms_on_slip_lists_through_the_byte_parallel_step(slip, full):
    """batched_step_ex with u_step / u_reset at slip_prob > 0 (step_kernel_swar<.., SLIPM = 3, ..>): the float64 decision against the
    nominal thresholds, with every group that holds a uniform within 2^-40 of one handed to the per-lane kernel's exact walk through
    the work list.  Three regimes per step: random uniforms (nothing listed), a mix, and EVERY lane on a threshold (everything
    listed: the whole batch goes through the one-workgroup tail) — eagerly and as a replayed graph (the list's count must come
    back to zero every time)."""
    n, steps = 8192, 12
    rng = np.random.default_rng(int(slip * 100) + full)
    b = SoccerBatch(n, 5, 4, slip, seed=3, autoreset=True, step_stats=full)
    o = Oracle(5, 4, slip, n=n, seed=3, autoreset=True)
    b.reset(); o.reset()
    c0 = (1 - slip) * (1 - slip); c1 = (1 - slip) * slip * 0.5
    on_thr = np.array([c0, np.nextafter(c0, 0), np.nextafter(c0, 1), c0 + c1, c0 + c1 + c1, c0 * 0.5, c0 * 0.25, c0 * 0.75, 0.0, 1.0 - 2.0 ** -53])
    A = b.alloc(n, np.int8); B = b.alloc(n, np.int8); U = b.alloc(n, np.float64); UR = b.alloc(n, np.float64)
    obs = b.alloc(n, np.uint16); rew = b.alloc(n, np.int8); te = b.alloc(n, np.uint8); tr = b.alloc(n, np.uint8)
    code = b.alloc(n, np.uint8) if full else None; fin = b.alloc(n, np.uint16) if full else None

    def one(k, captured=None):
        a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
        regime = k % 3
        u = rng.random(n) if regime == 0 else on_thr[rng.integers(0, len(on_thr), n)] if regime == 2 else \
            np.where(rng.random(n) < 0.3, on_thr[rng.integers(0, len(on_thr), n)], rng.random(n))
        ur = rng.random(n)
        A.upload(a[0]); B.upload(a[1]); U.upload(u); UR.upload(ur)
        if captured is None:
            b.step(A, B, obs=obs, reward=rew, terminated=te, truncated=tr, prob_code=code, final_obs=fin, u_step=U, u_reset=UR)
        else:
            b.graph_launch(captured, 1)
        c = o.step(a[0], a[1], u_step=u, u_reset=ur)
        np.testing.assert_array_equal(obs.download(), c["obs"], err_msg="obs, step %d regime %d" % (k, regime))
        np.testing.assert_array_equal(rew.download(), c["reward"]); np.testing.assert_array_equal(te.download(), c["terminated"])
        np.testing.assert_array_equal(tr.download(), c["truncated"])
        if full:
            np.testing.assert_array_equal(code.download(), c["prob_code"], err_msg="prob_code, step %d regime %d" % (k, regime))
            np.testing.assert_array_equal(fin.download(), c["final_obs"])
    for k in range(steps):
        one(k)
    b.graph_begin()
    b.step(A, B, obs=obs, reward=rew, terminated=te, truncated=tr, prob_code=code, final_obs=fin, u_step=U, u_reset=UR)
    g = b.graph_end()
    for k in range(steps):
        one(k, captured=g)
    _state_equal(b, o)
    if full:
        np.testing.assert_array_equal(b.stats()[0], o.hist)
    assert b.misuse() == 0 and b.tick == o.tick
    b.graph_destroy(g); b.close()
