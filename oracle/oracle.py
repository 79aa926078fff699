"""ctypes binding of the CPU oracle (oracle/soccer_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gym_soccer_littman94_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsoccer_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "soccer_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


class _State(C.Structure):
    _fields_ = [("row_a", C.c_void_p), ("col_a", C.c_void_p), ("row_b", C.c_void_p),
                ("col_b", C.c_void_p), ("poss", C.c_void_p), ("t", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.soc_oracle_create.restype = C.c_void_p
        L.soc_oracle_create.argtypes = [C.c_int, C.c_int, C.c_double]
        L.soc_oracle_destroy.argtypes = [C.c_void_p]
        L.soc_oracle_set_max_steps.argtypes = [C.c_void_p, C.c_int]
        for f in ("soc_oracle_ns", "soc_oracle_n_tuples", "soc_oracle_internal_width", "soc_oracle_n_isd"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.soc_oracle_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.soc_oracle_transitions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.soc_oracle_transitions.restype = C.c_int
        L.soc_oracle_dump_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.soc_oracle_dump_table.restype = C.c_int64
        L.soc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.soc_oracle_batched_reset.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_State), C.c_void_p,
                                               C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.soc_oracle_batched_reset.restype = C.c_int
        L.soc_oracle_batched_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(_State),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_uint64, C.c_uint64, C.c_uint64, C.c_int] + [C.c_void_p] * 8
        L.soc_oracle_batched_step.restype = C.c_int64
        L.soc_oracle_sample_actions.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64,
                                                C.c_void_p, C.c_void_p]
        L.soc_oracle_sample_actions_mixed.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    lib().soc_philox4x32_10(_p(c), _p(k), _p(o))
    return o


class Oracle:
    """The reference's env restated: tables built at construction, step = lookup + sample.

    Holds the batched SoA state of `n` lanes on the host with the product's semantics
    (include/soccer_hip.h): ticks, Philox streams, optional auto-reset.
    """

    def __init__(self, width=5, height=4, slip_prob=0.0, n=1, seed=0, lane_offset=0,
                 autoreset=False, max_steps=100):
        self.L = lib()
        self.h = self.L.soc_oracle_create(width, height, float(slip_prob))
        if not self.h:
            raise AssertionError("width must be >= 5 and height >= 4")
        self.L.soc_oracle_set_max_steps(self.h, max_steps)
        self.n = int(n); self.seed = int(seed); self.lane_offset = int(lane_offset)
        self.autoreset = bool(autoreset); self.tick = 0
        self.nS = self.L.soc_oracle_ns(self.h)
        self.n_tuples = self.L.soc_oracle_n_tuples(self.h)
        self.W = self.L.soc_oracle_internal_width(self.h); self.H = height
        self.n_isd = self.L.soc_oracle_n_isd(self.h)
        self.row_a = np.zeros(self.n, np.int8); self.col_a = np.zeros(self.n, np.int8)
        self.row_b = np.zeros(self.n, np.int8); self.col_b = np.zeros(self.n, np.int8)
        self.poss = np.full(self.n, 2, np.uint8)     # needs reset (:140)
        self.t = np.zeros(self.n, np.uint8)
        self.hist = np.zeros(3, np.uint64)
        self.misuse = 0
        # like the product, lanes are parked on the first ISD tuple until the first reset
        isd0 = self.tables()[3][0]
        self.row_a[:] = isd0[0]; self.col_a[:] = isd0[1]; self.row_b[:] = isd0[2]; self.col_b[:] = isd0[3]
        self.poss[:] = 2 | int(isd0[4])

    def __del__(self):
        try:
            if self.h:
                self.L.soc_oracle_destroy(self.h); self.h = None
        except Exception:
            pass

    def _state(self):
        return _State(*(a.ctypes.data for a in (self.row_a, self.col_a, self.row_b, self.col_b, self.poss, self.t)))

    def tables(self):
        lut = np.zeros(self.n_tuples, np.uint16); kind = np.zeros(self.n_tuples, np.uint8)
        gv = np.zeros(self.n_tuples, np.int8); isd = np.zeros((self.n_isd, 5), np.int8)
        isdp = np.zeros(self.n_isd, np.float64)
        self.L.soc_oracle_tables(self.h, _p(lut), _p(kind), _p(gv), _p(isd), _p(isdp))
        return lut, kind, gv, isd, isdp

    def transitions(self, st, aa, ab):
        s = np.asarray(st, np.int8); p = np.zeros(36); ns = np.zeros((36, 5), np.int8)
        r = np.zeros(36, np.int8); d = np.zeros(36, np.uint8)
        n = self.L.soc_oracle_transitions(self.h, _p(s), int(aa), int(ab), _p(p), _p(ns), _p(r), _p(d))
        if n < 0:
            raise KeyError(tuple(int(x) for x in st))
        return p[:n], ns[:n], r[:n], d[:n]

    def dump_table(self):
        """The complete transition relation in the canonical order of tests/golden/make_golden.py::table_digest:
        rows int8[n, 15] (xa,ya,xb,yb,p, aa,ab, k, next tuple, reward, done), prob float64[n]."""
        n = self.L.soc_oracle_dump_table(self.h, None, None, 0)
        rows = np.zeros((n, 15), np.int8); prob = np.zeros(n, np.float64)
        assert self.L.soc_oracle_dump_table(self.h, _p(rows), _p(prob), n) == n
        return rows, prob

    def set_state(self, row_a, col_a, row_b, col_b, poss, t=None, needs_reset=None):
        self.row_a[:] = row_a; self.col_a[:] = col_a; self.row_b[:] = row_b; self.col_b[:] = col_b
        nr = (self.poss >> 1) & 1 if needs_reset is None else np.asarray(needs_reset, np.uint8)
        self.poss[:] = (np.asarray(poss, np.uint8) & 1) | (nr << 1)
        if t is not None:
            self.t[:] = t

    def reseed(self, seed):
        self.seed = int(seed); self.tick = 0

    def reset(self, mask=None, u_reset=None):
        obs = np.zeros(self.n, np.uint16)
        st = self._state()
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        u = None if u_reset is None else np.ascontiguousarray(u_reset, np.float64)
        self.L.soc_oracle_batched_reset(self.h, self.n, C.byref(st), _p(m), _p(u),
                                        self.seed, self.lane_offset, self.tick, _p(obs))
        self.tick += 1
        return obs

    def step(self, act_a, act_b, u_step=None, u_reset=None):
        n = self.n
        a = np.ascontiguousarray(act_a, np.int8); b = np.ascontiguousarray(act_b, np.int8)
        us = None if u_step is None else np.ascontiguousarray(u_step, np.float64)
        ur = None if u_reset is None else np.ascontiguousarray(u_reset, np.float64)
        out = dict(obs=np.zeros(n, np.uint16), reward=np.zeros(n, np.int8),
                   terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8),
                   prob=np.zeros(n, np.float64), prob_code=np.zeros(n, np.uint8),
                   final_obs=np.zeros(n, np.uint16))
        st = self._state()
        self.misuse += self.L.soc_oracle_batched_step(
            self.h, n, C.byref(st), _p(a), _p(b), _p(us), _p(ur), self.seed, self.lane_offset,
            self.tick, int(self.autoreset), _p(out["obs"]), _p(out["reward"]), _p(out["terminated"]),
            _p(out["truncated"]), _p(out["prob"]), _p(out["prob_code"]), _p(out["final_obs"]),
            _p(self.hist))
        self.tick += 1
        return out

    def sample_actions(self, tick=None):
        a = np.zeros(self.n, np.int8); b = np.zeros(self.n, np.int8)
        self.L.soc_oracle_sample_actions(self.n, self.seed, self.lane_offset,
                                         self.tick if tick is None else tick, _p(a), _p(b))
        return a, b

    def sample_actions_mixed(self, obs_now, mix_a=None, mix_b=None):
        a = np.zeros(self.n, np.int8); b = np.zeros(self.n, np.int8)
        o = np.ascontiguousarray(obs_now, np.uint16)
        ma = None if mix_a is None else np.ascontiguousarray(mix_a, np.uint16)
        mb = None if mix_b is None else np.ascontiguousarray(mix_b, np.uint16)
        self.L.soc_oracle_sample_actions_mixed(self.n, self.seed, self.lane_offset, self.tick, _p(o), _p(ma), _p(mb), _p(a), _p(b))
        return a, b


# -------------------------------------------------------------------------------------------------
# planners (reference gym_soccer/utils/planners.py:4-18) — restated over the oracle's transition lists
# -------------------------------------------------------------------------------------------------
def single_agent_lists(orc, learner, policy):
    """P[s][a] of a single-agent env as the reference's constructor builds it (:167-293): the fixed side
    plays policy[s]; learner B's reward is flipped (:243-244); goal tuples all write index 0."""
    lut, kind, gv, isd, isdp = orc.tables()
    H, W = orc.H, orc.W
    P = {}
    for f in np.flatnonzero(kind != 0).tolist():
        p_ = f & 1; r = f >> 1
        yb = r % W; r //= W; xb = r % H; r //= H; ya = r % W; xa = r // W
        s = 0 if kind[f] == 2 else int(lut[f])
        P[s] = {}
        for a in range(5):
            aa, ab = (a, int(policy[s])) if learner == "player_a" else (int(policy[s]), a)
            ps, ns, rs, ds = orc.transitions((xa, ya, xb, yb, p_), aa, ab)
            lst = []
            for k in range(len(ps)):
                nf = ((((int(ns[k][0]) * W + int(ns[k][1])) * H + int(ns[k][2])) * W + int(ns[k][3])) << 1) | int(ns[k][4])
                rr = float(rs[k])
                if learner == "player_b":
                    rr = -1 * rr
                lst.append((float(ps[k]), 0 if kind[nf] == 2 else int(lut[nf]), rr, bool(ds[k])))
            P[s][a] = lst
    return P


def value_iteration(P, nS, theta=1e-10, discount_factor=0.99, max_iterations=1000000):
    """planners.py:4-18: synchronous sweeps, python-float accumulation in list order."""
    V = [0.0] * nS
    cc = 0
    while True:
        cc += 1
        Q = [[0.0] * 5 for _ in range(nS)]
        for s in range(nS):
            for a in range(5):
                q = 0.0
                for prob, ns, reward, done in P[s][a]:
                    q += prob * (reward + discount_factor * V[ns] * (not done))
                Q[s][a] = q
        newV = [max(row) for row in Q]
        if max(abs(V[s] - newV[s]) for s in range(nS)) < theta or cc >= max_iterations:
            break
        V = newV
    Qn = np.array(Q)
    return np.argmax(Qn, axis=1), np.array(V), Qn, cc


def single_agent_mats(orc, learner, policy):
    """Pmat[s, ns, a] / Rmat[s, a] as the reference's constructor accumulates them (:280-291): per reachable
    tuple in loop order, Rmat[s][a] = 0 then += p * r, Pmat[s][ns][a] += p (index 0 collects every goal tuple)."""
    lut, kind, gv, isd, isdp = orc.tables()
    H, W = orc.H, orc.W
    nS = orc.nS
    Pmat = np.zeros([nS, nS, 5]); Rmat = np.zeros([nS, 5])
    for f in np.flatnonzero(kind != 0).tolist():
        p_ = f & 1; r = f >> 1
        yb = r % W; r //= W; xb = r % H; r //= H; ya = r % W; xa = r // W
        s = 0 if kind[f] == 2 else int(lut[f])
        for a in range(5):
            aa, ab = (a, int(policy[s])) if learner == "player_a" else (int(policy[s]), a)
            ps, ns, rs, ds = orc.transitions((xa, ya, xb, yb, p_), aa, ab)
            Rmat[s][a] = 0
            for k in range(len(ps)):
                nf = ((((int(ns[k][0]) * W + int(ns[k][1])) * H + int(ns[k][2])) * W + int(ns[k][3])) << 1) | int(ns[k][4])
                rr = float(rs[k])
                if learner == "player_b":
                    rr = -1 * rr
                Pmat[s][0 if kind[nf] == 2 else int(lut[nf])][a] += float(ps[k])
                Rmat[s][a] += float(ps[k]) * rr
    return Pmat, Rmat


def policy_evaluation(pi, P, nS, theta, discount_factor):
    """planners.py:20-31."""
    prev_V = [0.0] * nS
    sweeps = 0
    while True:
        V = [0.0] * nS
        for s in range(nS):
            v = 0.0
            for prob, ns, reward, done in P[s][int(pi[s])]:
                v += prob * (reward + discount_factor * prev_V[ns] * (not done))
            V[s] = v
        sweeps += 1
        if max(abs(prev_V[s] - V[s]) for s in range(nS)) < theta:
            break
        prev_V = V
    return np.array(V), sweeps


def policy_improvement(V, P, nS, discount_factor):
    """planners.py:33-41."""
    Q = np.zeros((nS, 5))
    for s in range(nS):
        for a in range(5):
            q = 0.0
            for prob, ns, reward, done in P[s][a]:
                q += prob * (reward + discount_factor * float(V[ns]) * (not done))
            Q[s][a] = q
    return np.argmax(Q, axis=1), Q


def policy_iteration(P, nS, pi0, theta, discount_factor):
    """planners.py:43-53 with the initial draw passed in."""
    cc = 0
    pi = np.asarray(pi0).copy()
    while True:
        old_pi = pi.copy()
        V, _ = policy_evaluation(pi, P, nS, theta, discount_factor)
        pi, Q = policy_improvement(V, P, nS, discount_factor)
        cc += 1
        if np.all(old_pi == pi):
            break
    return pi, V, Q, cc


def policy_eval_dense(Pmat, Rmat, policy, theta, discount_factor, k=10000000, init=None):
    """planners.py:55-70."""
    nS = Rmat.shape[0]
    v = np.zeros(nS) if init is None else init
    cc = 0
    for _ in range(k):
        value_fc = np.zeros(nS)
        for s in range(nS):
            r_pi = np.dot(policy[s, :], Rmat[s, :])
            pv = np.dot(Pmat[s, :, :].T, v)
            value_fc[s] = r_pi + discount_factor * np.dot(pv, policy[s, :])
        delta = np.max(np.abs(value_fc - v))
        v[:] = value_fc
        cc += 1
        if delta < theta:
            break
    return v, cc


def modified_policy_iteration(Pmat, Rmat, k, theta, discount_factor):
    """planners.py:73-87."""
    nS = Rmat.shape[0]
    v = np.zeros(nS)
    threshold = (theta * (1 - discount_factor)) / (2 * discount_factor)
    counter = 0
    while True:
        q = np.zeros([nS, 5])
        for a in range(5):
            q[:, a] = Rmat[:, a] + discount_factor * np.dot(Pmat[:, :, a], v)
        greedy_v = np.max(q, -1)
        best = np.argmax(q, -1)
        if np.max(np.abs(v - greedy_v)) <= threshold:
            return best, greedy_v, q, counter
        v, _ = policy_eval_dense(Pmat, Rmat, np.eye(5)[best], theta, discount_factor, k=k, init=greedy_v)
        counter += 1
