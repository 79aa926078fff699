"""The byte-parallel step of the hot kernels (csrc/soccer_swar.hpp: four lanes per dword, no rule tables) against the
oracle, on the CPU, EXHAUSTIVELY: every reachable tuple (live and goal) x 25 joint actions x 4 outcome draws x 4 reset
draws, at timesteps around the truncation, frozen and not, with and without auto-reset, on every pitch the reference
parametrises (tests/test_general.py:5-11) — 5x4 ... 11x7.  The header compiles for the host with the four GPU builtins
it uses restated in C++ (tests/host/swar_host.cpp), so this is the code the kernels run.  The oracle itself is pinned
to the reference by tests/test_oracle_golden.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("swar") / "libswar_host.so")
    # SWAR_HOST_SANITIZE=1: the same tests with UndefinedBehaviorSanitizer in the host build of the byte-parallel rules
    # (shifts, signed overflow, misaligned access ...; any report aborts the process) — sanitizers run on the CPU build only
    san = ["-fsanitize=undefined", "-fno-sanitize-recover=all", "-g"] if os.environ.get("SWAR_HOST_SANITIZE") else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror"] + san + ["-o", so,
                           os.path.join(ROOT, "tests", "host", "swar_host.cpp")])
    L = C.CDLL(so)
    L.swar_step_host.restype = C.c_int
    L.swar_step_host.argtypes = [C.c_int] * 6 + [C.c_long] + [C.c_void_p] * 9 + [C.c_double] + [C.c_void_p] * 9
    L.swar_draws_host.argtypes = [C.c_void_p, C.c_void_p]
    L.swar_slip_tables.argtypes = [C.c_double] + [C.c_void_p] * 4
    L.swar_set_slip_select.argtypes = [C.c_int]
    L.swar_reset_host.restype = C.c_int
    L.swar_reset_host.argtypes = [C.c_int, C.c_int, C.c_long] + [C.c_void_p] * 9
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _tuples(o, kinds):
    lut, kind, gv, isd, isdp = o.tables()
    f = np.flatnonzero(np.isin(kind, kinds))
    p = f & 1; r = f >> 1
    cb = r % o.W; r //= o.W; rb = r % o.H; r //= o.H; ca = r % o.W; ra = r // o.W
    return np.stack([ra, ca, rb, cb, p], 1).astype(np.int64)


def _run_swar(L, w, h, max_steps, autoreset, general, full, st, t, need, aa, ab, words, slip=0.0):
    n = len(t)
    ra, ca, rb, cb = (np.ascontiguousarray(st[:, k], np.uint8) for k in range(4))
    ps = np.ascontiguousarray(st[:, 4] | (need << 1), np.uint8)
    tt = np.ascontiguousarray(t, np.uint8)
    out = dict(obs=np.zeros(n, np.uint16), final_obs=np.zeros(n, np.uint16), reward=np.zeros(n, np.uint8),
               terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8), prob_code=np.zeros(n, np.uint8),
               finished=np.zeros(n, np.uint8), frozen=np.zeros(n, np.uint8), bad=np.zeros(n // 4, np.uint8))
    rc = L.swar_step_host(w, h, max_steps, int(autoreset), int(general), int(full), n,
                          _p(ra), _p(ca), _p(rb), _p(cb), _p(ps), _p(tt),
                          _p(np.ascontiguousarray(aa, np.uint8)), _p(np.ascontiguousarray(ab, np.uint8)),
                          _p(np.ascontiguousarray(words, np.uint32)), float(slip),
                          _p(out["obs"]), _p(out["final_obs"]), _p(out["reward"]), _p(out["terminated"]),
                          _p(out["truncated"]), _p(out["prob_code"]), _p(out["finished"]), _p(out["frozen"]), _p(out["bad"]))
    if rc == -4:
        pytest.skip("slip %r has no bucket table (its thresholds crowd a bucket): the kernels compare one by one" % slip)
    assert rc == 0
    out["reward"] = out["reward"].view(np.int8)
    out["state"] = (ra, ca, rb, cb, ps, tt)
    return out


def _grid(tup, t_values, need_values, rng):
    """every tuple x 25 joint actions x 4 outcome draws x 4 reset draws, timestep / frozen flag cycled"""
    nt = len(tup)
    idx = np.arange(nt * 25 * 16)
    ti = idx // 400; r = idx % 400
    aa = r // 80; r %= 80
    ab = r // 16; r %= 16
    top2 = r // 4; reset2 = r % 4
    n = len(idx)
    pad = (-n) % 4
    if pad:
        ti = np.concatenate([ti, ti[:pad]]); aa = np.concatenate([aa, aa[:pad]]); ab = np.concatenate([ab, ab[:pad]])
        top2 = np.concatenate([top2, top2[:pad]]); reset2 = np.concatenate([reset2, reset2[:pad]])
        n += pad
    mid = rng.integers(0, 1 << 28, size=n, dtype=np.uint32)
    words = (top2.astype(np.uint32) << 30) | (mid << 2) | reset2.astype(np.uint32)
    t = np.asarray(t_values)[rng.integers(0, len(t_values), size=n)]
    need = np.asarray(need_values)[rng.integers(0, len(need_values), size=n)]
    return tup[ti], t, need.astype(np.int64), aa, ab, words


def _oracle_step(w, h, max_steps, autoreset, st, t, need, aa, ab, words, slip=0.0):
    n = len(t)
    o = Oracle(w, h, slip, n=n, autoreset=autoreset, max_steps=max_steps)
    o.set_state(st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], t=t, needs_reset=need)
    # the uniforms of one word per lane (include/soccer_hip.h): u = (m + 1/2) * 2^-b
    c = o.step(aa, ab, u_step=((words >> 2).astype(np.float64) + 0.5) * 2.0 ** -30, u_reset=((words & 3).astype(np.float64) + 0.5) * 0.25)
    c["state"] = (o.row_a.view(np.uint8), o.col_a.view(np.uint8), o.row_b.view(np.uint8), o.col_b.view(np.uint8), o.poss, o.t)
    return c


def _compare(got, exp, need, full):
    for k in ("obs", "reward", "terminated", "truncated") + (("final_obs", "prob_code") if full else ()):
        bad = np.flatnonzero(got[k] != exp[k])
        assert bad.size == 0, "%s differs on %d lanes, first %d: got %s expected %s" % (k, bad.size, bad[0], got[k][bad[0]], exp[k][bad[0]])
    for k, name in enumerate(("row_a", "col_a", "row_b", "col_b", "poss|needs_reset<<1", "t")):
        bad = np.flatnonzero(got["state"][k] != exp["state"][k])
        assert bad.size == 0, "state %s differs on %d lanes, first %d" % (name, bad.size, bad[0])
    fin = ((exp["terminated"] | exp["truncated"]) != 0) & (need == 0)
    np.testing.assert_array_equal(got["finished"], fin.astype(np.uint8))
    np.testing.assert_array_equal(got["frozen"], need.astype(np.uint8))
    assert not got["bad"].any()


PITCHES = [(5, 4), (6, 4), (7, 5), (9, 6), (11, 7)]


@pytest.fixture(params=["tables", "arithmetic"])
def geometry(request, host):
    """both instantiations of the pitch geometry: byte tables (small pitches; larger ones take the arithmetic anyway)
    and arithmetic forced on every pitch"""
    host.swar_set_geo(-1 if request.param == "tables" else 0)
    yield request.param
    host.swar_set_geo(-1)


@pytest.mark.parametrize("w,h", PITCHES)
@pytest.mark.parametrize("autoreset", [True, False])
def test_general_step_every_tuple_action_and_draw(host, geometry, w, h, autoreset):
    if geometry == "arithmetic" and (w, h) not in ((5, 4), (6, 4)):
        pytest.skip("already arithmetic")
    rng = np.random.default_rng(w * 100 + h)
    o = Oracle(w, h, 0.0, n=1)
    tup = _tuples(o, [1, 2])                                   # live and goal tuples
    st, t, need, aa, ab, words = _grid(tup, [0, 1, 57, 98, 99, 100], [0, 0, 0, 1], rng)
    t = np.where(need == 1, t, np.minimum(t, 99))              # a lane that is not frozen has t < max_steps
    got = _run_swar(host, w, h, 100, autoreset, True, True, st, t, need, aa, ab, words)
    exp = _oracle_step(w, h, 100, autoreset, st, t, need, aa, ab, words)
    _compare(got, exp, need, True)
    lean = _run_swar(host, w, h, 100, autoreset, True, False, st, t, need, aa, ab, words)
    _compare(lean, exp, need, False)


@pytest.mark.parametrize("w,h", PITCHES)
def test_steady_state_step_every_live_tuple_action_and_draw(host, geometry, w, h):
    if geometry == "arithmetic" and (w, h) not in ((5, 4), (6, 4)):
        pytest.skip("already arithmetic")
    """the instantiation of an auto-resetting handle whose lanes have all been reset: no frozen / goal-tuple code"""
    rng = np.random.default_rng(w * 100 + h + 1)
    o = Oracle(w, h, 0.0, n=1)
    tup = _tuples(o, [1])
    st, t, need, aa, ab, words = _grid(tup, [0, 3, 98, 99], [0], rng)
    exp = _oracle_step(w, h, 100, True, st, t, need, aa, ab, words)
    for full in (True, False):
        got = _run_swar(host, w, h, 100, True, False, full, st, t, need, aa, ab, words)
        _compare(got, exp, need, full)


def test_other_episode_lengths(host):
    rng = np.random.default_rng(5)
    o = Oracle(5, 4, 0.0, n=1)
    tup = _tuples(o, [1, 2])
    for ms in (1, 2, 17, 127):
        st, t, need, aa, ab, words = _grid(tup, [0, max(ms - 2, 0), ms - 1, ms], [0, 0, 1], rng)
        t = np.where(need == 1, t, np.minimum(t, ms - 1))
        got = _run_swar(host, 5, 4, ms, True, True, True, st, t, need, aa, ab, words)
        exp = _oracle_step(5, 4, ms, True, st, t, need, aa, ab, words)
        _compare(got, exp, need, True)


def test_actions_outside_0_4_are_reported_and_never_leave_the_pitch(host):
    rng = np.random.default_rng(9)
    o = Oracle(5, 4, 0.0, n=1)
    tup = _tuples(o, [1])
    n = 4 * 20000
    st = tup[rng.integers(0, len(tup), size=n)]
    aa = rng.integers(0, 5, size=n).astype(np.uint8); ab = rng.integers(0, 5, size=n).astype(np.uint8)
    bad_lane = np.arange(0, n, 8) + rng.integers(0, 8, size=n // 8)           # one per two groups
    aa_bad = aa.copy(); ab_bad = ab.copy()
    vals = np.array([5, 6, 7, 8, 127, 128, 255, 0x85], np.uint8)
    which = rng.integers(0, 2, size=len(bad_lane)).astype(bool)
    aa_bad[bad_lane[which]] = vals[rng.integers(0, len(vals), size=which.sum())]
    ab_bad[bad_lane[~which]] = vals[rng.integers(0, len(vals), size=(~which).sum())]
    words = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    t = rng.integers(0, 99, size=n); need = np.zeros(n, np.int64)
    got = _run_swar(host, 5, 4, 100, True, True, False, st, t, need, aa_bad, ab_bad, words)
    flagged = np.zeros(n // 4, bool); flagged[bad_lane // 4] = True
    np.testing.assert_array_equal(got["bad"].astype(bool), flagged)
    # every lane still holds a reachable tuple, and the lanes of clean groups stepped exactly as the oracle does
    lut, kind, *_ = o.tables()
    ra, ca, rb, cb, ps, tt = (x.astype(np.int64) for x in got["state"])
    assert (ra < 4).all() and (rb < 4).all() and (ca < 7).all() and (cb < 7).all()
    assert (kind[(((ra * 7 + ca) * 4 + rb) * 7 + cb) * 2 + (ps & 1)] == 1).all()      # auto-reset on: live tuples only
    exp = _oracle_step(5, 4, 100, True, st, t, need, aa, ab, words)
    clean = np.repeat(~flagged, 4)
    for k in ("obs", "reward", "terminated", "truncated"):
        np.testing.assert_array_equal(got[k][clean], exp[k][clean])


# ---- slip_prob > 0: per-lane combination / quarter selection (integer thresholds) + the byte-parallel step ----------------
def _slip_words(slip, n, rng):
    """draws that cover every combination and every quarter of each: uniform words, plus words sitting exactly on, one
    below and one above every scaled threshold of the handle (the integer decision must flip exactly there)"""
    s = slip
    c = [(1 - s) * (1 - s), (1 - s) * s * 0.5, (1 - s) * s * 0.5, s * (1 - s) * 0.5, s * (1 - s) * 0.5] + [s * s * 0.25] * 4
    edges = []
    acc = 0.0
    for wgt in c:
        if wgt == 0.0:
            continue
        for q in (0.25, 0.5, 0.75, 1.0):
            edges.append(acc + wgt * q)
        acc += wgt
    m_edge = np.array([int(np.ceil(e * 2 ** 30 - 0.5)) for e in edges if e < 1.0], dtype=np.int64)
    special = np.concatenate([m_edge - 1, m_edge, m_edge + 1, [0, 1, 2 ** 30 - 1]])
    special = special[(special >= 0) & (special < 2 ** 30)]
    m = rng.integers(0, 1 << 30, size=n, dtype=np.int64)
    pick = rng.random(n) < 0.25
    m[pick] = special[rng.integers(0, len(special), size=int(pick.sum()))]
    return ((m << 2) | rng.integers(0, 4, size=n)).astype(np.uint32)


@pytest.fixture(params=["one_by_one", "table", "step_table"])
def selection(request, host):
    """the forms of the per-lane slip selection the kernels have: the thresholds compared one by one (slip_select4), the
    bucket table over the draw's top 14 bits + one exact compare (slip_select4_lut, what the rollout takes whenever the
    slip's thresholds do not crowd a bucket) and the single step's 10-bit table + two exact compares"""
    host.swar_set_slip_select({"one_by_one": 1, "table": 2, "step_table": 3}[request.param])
    yield request.param
    host.swar_set_slip_select(1)


@pytest.mark.parametrize("w,h,slip", [(5, 4, 0.2), (5, 4, 0.5), (5, 4, 1.0), (5, 4, 0.3), (5, 4, 0.05), (7, 5, 0.3), (11, 7, 0.2),
                                      (5, 4, 0.1), (5, 4, 0.9), (5, 4, 0.15), (6, 4, 0.4), (5, 4, 2.0 / 3.0), (5, 4, 0.6)])
def test_slip_step_every_tuple_and_action_with_draws_on_every_threshold(host, selection, w, h, slip):
    rng = np.random.default_rng(int(slip * 100) + w)
    o = Oracle(w, h, slip, n=1)
    tup = _tuples(o, [1, 2])
    reps = 6 if w == 5 else 2
    idx = np.arange(len(tup) * 25 * reps)
    ti = idx // (25 * reps); r = idx % (25 * reps)
    aa = (r // reps) // 5; ab = (r // reps) % 5
    n = len(idx) - len(idx) % 4
    ti, aa, ab = ti[:n], aa[:n], ab[:n]
    words = _slip_words(slip, n, rng)
    t = np.asarray([0, 40, 98, 99, 100])[rng.integers(0, 5, size=n)]
    need = (rng.random(n) < 0.2).astype(np.int64)
    t = np.where(need == 1, t, np.minimum(t, 99))
    st = tup[ti]
    for autoreset in (True, False):
        got = _run_swar(host, w, h, 100, autoreset, True, True, st, t, need, aa, ab, words, slip=slip)
        exp = _oracle_step(w, h, 100, autoreset, st, t, need, aa, ab, words, slip=slip)
        _compare(got, exp, need, True)
    # the lean, steady-state instantiation on live tuples
    live = _tuples(o, [1])
    st = live[rng.integers(0, len(live), size=n)]; need0 = np.zeros(n, np.int64); t0 = np.minimum(t, 99)
    got = _run_swar(host, w, h, 100, True, False, False, st, t0, need0, aa, ab, words, slip=slip)
    exp = _oracle_step(w, h, 100, True, st, t0, need0, aa, ab, words, slip=slip)
    _compare(got, exp, need0, False)


SLIPS = [0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 1.0 / 3.0, 0.33, 0.4, 0.5, 0.6, 2.0 / 3.0, 0.75, 0.9, 0.999, 1.0, 1e-9, 0.123]


@pytest.mark.parametrize("slip", SLIPS)
def test_integer_slip_thresholds_are_the_float64_cumsum_for_every_list_shape(host, slip):
    """soccer_slip.hpp accepts the integer slip decision only when, at every entry position, c = ceil(t * 2^30 - 1/2) is the
    same for the running sum t of EVERY list shape.  Checked here independently: random shapes (each combination
    contributing 1, 2 or 4 entries), the running sums by np.cumsum — the reference's own categorical_sample arithmetic —
    and c with exact rational arithmetic.  Decimal slips such as 0.1 / 0.15 / 0.4 / 2/3, whose mathematically dyadic
    thresholds needed a float64 walk inside the kernels under round 2's u = m * 2^-30, are plain integer handles now."""
    from fractions import Fraction
    import math
    cb = np.zeros(9, np.uint32); sub = np.zeros(36, np.uint32); flags = np.zeros(8, np.uint32); w4 = np.zeros(4, np.float64)
    host.swar_slip_tables(float(slip), _p(cb), _p(sub), _p(flags), _p(w4))
    assert flags[0] == 1 and flags[1] == 1, "slip %r should take the integer decision on the byte-parallel path" % slip
    s = slip
    c = [(1 - s) * (1 - s), (1 - s) * s * 0.5, (1 - s) * s * 0.5, s * (1 - s) * 0.5, s * (1 - s) * 0.5] + [s * s * 0.25] * 4   # :209-223
    np.testing.assert_array_equal(w4, [c[0], c[1], c[3], c[5]])
    active = [x for x in c if x != 0.0]                                # :226-227
    assert len(active) == flags[2]
    rng = np.random.default_rng(int(slip * 1e6) % 9973)
    def c_of(t):
        return min(max(math.ceil(Fraction(float(t)) * 2 ** 30 - Fraction(1, 2)), 0), 2 ** 30)
    for _ in range(400):
        shape = rng.choice([1, 2, 4], size=len(active))
        plist = [wgt * {1: 1.0, 2: 0.5, 4: 0.25}[int(n)] for wgt, n in zip(active, shape) for _k in range(int(n))]   # :241
        run = np.cumsum(np.asarray(plist))                             # gym's categorical_sample: cumsum(asarray(p)) > u
        pos = 0
        for i, n in enumerate(shape):
            ends = [c_of(t) for t in run[pos:pos + n]]
            pos += n
            assert ends[-1] == cb[i]
            if n == 2:
                assert ends[0] == sub[4 * i] == sub[4 * i + 2]
            if n == 4:
                assert ends[:3] == [int(x) for x in sub[4 * i + 1:4 * i + 4]]
    assert cb[len(active) - 1] == 2 ** 30                              # no draw falls beyond the last entry
    # the table form exists unless two of the 4 * nb thresholds share one 2^16-wide bucket of the draw (tiny s: the four
    # s^2 / 4 combinations; s near 1: the (1 - s)^2 one)
    thr = np.sort(np.concatenate([np.stack([sub[1::4], sub[2::4], sub[3::4], cb], 1)[:len(active)].ravel()]).astype(np.int64))
    inside = np.bincount((thr[(thr > 0) & (thr < 2 ** 30) & (thr % 2 ** 16 != 0)] >> 16), minlength=16384)
    assert bool(flags[4]) == bool(inside.max() <= 1), (slip, flags[4], inside.max())
    if slip in (0.1, 0.2, 0.5, 1.0 / 3.0):
        assert flags[4] == 1
    # the single step's table: 2^20-wide buckets, up to two thresholds inside one
    inside = np.bincount((thr[(thr > 0) & (thr < 2 ** 30) & (thr % 2 ** 20 != 0)] >> 20), minlength=1024)
    assert bool(flags[5]) == bool(inside.max() <= 2), (slip, flags[5], inside.max())
    if slip in (0.1, 0.2, 0.5, 1.0 / 3.0, 0.9):
        assert flags[5] == 1


def test_eight_ticks_share_one_block_at_slip_zero(host):
    """slip_prob == 0 (include/soccer_hip.h, ABI 3): tick k takes nibble (k & 7) ^ 1 of the lane's word of block k >> 3;
    quarter draw = the nibble's two high bits, reset draw = its two low bits.  Both extraction forms of the kernels
    (rotate + pack for single steps, transposed block + pair walk for the rollout) against that definition."""
    rng = np.random.default_rng(3)
    for _ in range(200):
        w = rng.integers(0, 1 << 32, size=4, dtype=np.uint64).astype(np.uint32)
        out = np.zeros(64, np.uint8)
        host.swar_draws_host(_p(w), _p(out))
        out = out.reshape(8, 2, 4)
        for t in range(8):
            nib = (w >> np.uint32(4 * (t ^ 1))) & np.uint32(15)
            exp = (nib >> 2) | ((nib & 3) << 4)
            np.testing.assert_array_equal(out[t, 0], exp)
            np.testing.assert_array_equal(out[t, 1], exp)


@pytest.mark.parametrize("w,h", PITCHES)
def test_reset_all_lanes_and_masked(host, w, h):
    """batched_reset's byte-parallel form: every lane, and masked (mask bytes 0 / 1 / 255 / 128) over every reachable tuple
    incl. goal tuples and frozen lanes — state, needs_reset, timestep and the observation of every lane vs the oracle"""
    rng = np.random.default_rng(w)
    o = Oracle(w, h, 0.0, n=1)
    tup = _tuples(o, [1, 2])
    reps = 8
    st = np.repeat(tup, reps, axis=0)
    n = len(st) - len(st) % 4
    st = st[:n]
    words = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    words = (words & ~np.uint32(3)) | (np.arange(n) % 4).astype(np.uint32)
    t = rng.integers(0, 101, size=n); need = rng.integers(0, 2, size=n)
    for mask in (None, np.array([0, 1, 255, 128, 0, 0, 2, 0], np.uint8)[rng.integers(0, 8, size=n)]):
        oo = Oracle(w, h, 0.0, n=n)
        oo.set_state(st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], t=t, needs_reset=need)
        exp = oo.reset(mask=mask, u_reset=((words & 3).astype(np.float64) + 0.5) * 0.25)
        ra, ca, rb, cb = (np.ascontiguousarray(st[:, k], np.uint8) for k in range(4))
        ps = np.ascontiguousarray(st[:, 4] | (need << 1), np.uint8); tt = np.ascontiguousarray(t, np.uint8)
        obs = np.zeros(n, np.uint16)
        assert host.swar_reset_host(w, h, n, _p(ra), _p(ca), _p(rb), _p(cb), _p(ps), _p(tt), _p(mask), _p(words), _p(obs)) == 0
        np.testing.assert_array_equal(obs, exp)
        for got, ref in ((ra, oo.row_a), (ca, oo.col_a), (rb, oo.row_b), (cb, oo.col_b), (ps, oo.poss), (tt, oo.t)):
            np.testing.assert_array_equal(got, ref.view(np.uint8))
