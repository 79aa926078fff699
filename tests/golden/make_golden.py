#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the *real* reference.

Runs ONLY in the build container, where the upstream reference is mounted
read-only at /root/reference.  It is never run on the GPU box and nothing it
imports travels: the fixtures it writes are plain data (inputs + expected
outputs) in .npz files.

The reference's one missing dependency is `gym` (setup.py:14, not installed,
no network).  Four symbols are used (soccer_simultaneous_env.py:2-3,
gym_soccer/__init__.py:1); this script writes a throw-away stand-in for those
four symbols into a temp dir (our own ~15 lines, including the one-line
restatement of gym 0.26.2's `categorical_sample`:
`argmax(cumsum(asarray(p)) > np_random.random())`) and puts it on sys.path.

What is dumped
  table_{W}x{H}_s{slip}.npz   complete transition relation (P_readable) with
                              list order and float64 probabilities, state
                              classification, observation indices, ISD
  replay_{W}x{H}_s{slip}.npz  (state, t, joint action, u) vectors pushed through
                              the real step() with an injected uniform
  reset_{W}x{H}.npz           reset() with injected uniforms
  traj_5x4_s{slip}_seed{S}.npz  reset(seed=S) + MT19937-driven episodes through
                              the real reset()/step(), with the uniforms the
                              env drew recorded alongside
  single_5x4_s{slip}_{agent}.npz  single-agent (fixed-opponent) transition table
  vi_5x4_s{slip}_{agent}_vs_{opp}.npz  the reference's value_iteration (utils/planners.py) on its own tables

  digest_{W}x{H}_s{slip}.npz  for tables too large to commit (11x7 with slip: 3.4 M rows): the state
                              classification + SHA-256 of the canonical full-table dump (table_digest below)
                              + one 32-bit digest per state tuple to localise a mismatch
  traj_5x4_s{slip}_seed0_10k.npz  BASELINE config 1 at its stated length (10 000 steps)

Usage:  python tests/golden/make_golden.py                  (everything below 11x7)
        python tests/golden/make_golden.py one 11 7 0.0 4000   (one pitch: table + replay + reset)
        python tests/golden/make_golden.py digest 11 7 0.2 4000   (one pitch: table DIGEST + replay + reset)
        python tests/golden/make_golden.py traj10k              (the two 10 000-step trajectories)
"""
import json
import os
import sys
import tempfile
import textwrap
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_gym_stand_in():
    d = tempfile.mkdtemp(prefix="gymstandin_")
    files = {
        "gym/__init__.py": "from . import spaces\n",
        "gym/spaces.py": textwrap.dedent("""
            class Discrete:
                def __init__(self, n):
                    self.n = int(n)
            class Dict(dict):
                pass
            """),
        "gym/envs/__init__.py": "",
        "gym/envs/registration.py": "def register(*a, **k):\n    pass\n",
        "gym/envs/toy_text/__init__.py": "",
        "gym/envs/toy_text/utils.py": textwrap.dedent("""
            import numpy as np
            def categorical_sample(prob_n, np_random):
                prob_n = np.asarray(prob_n)
                csprob_n = np.cumsum(prob_n)
                return np.argmax(csprob_n > np_random.random())
            """),
    }
    for rel, src in files.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(src)
    sys.path.insert(0, d)
    sys.path.insert(1, REF)
    sys.dont_write_bytecode = True


class _FixedU:
    """np_random replacement returning an injected uniform."""
    def __init__(self, u):
        self.u = u
    def random(self):
        return self.u


class _RecordingRS:
    """Wraps a RandomState and records every uniform the env draws."""
    def __init__(self, rs):
        self.rs = rs
        self.drawn = []
    def seed(self, s):
        self.rs.seed(s)
    def random(self):
        u = self.rs.random()
        self.drawn.append(u)
        return u


def slip_tag(s):
    return ("%g" % s).replace(".", "p")


def table_digest(rows, prob):
    """Canonical digest of a full transition table.  rows int8[n, 15] = xa,ya,xb,yb,p, aa,ab, k, nxa,nya,nxb,nyb,np, reward, done
    in ascending (flat tuple index, aa, ab) order with k ascending (the reference's own P_readable iteration order), prob
    float64[n].  Returns (sha256 hex of rows bytes + little-endian prob bytes, uint32 digest per state tuple in order of
    appearance: the first four bytes of the SHA-256 of that tuple's rows + probs)."""
    import hashlib
    rows = np.ascontiguousarray(rows, dtype=np.int8); prob = np.ascontiguousarray(prob, dtype="<f8")
    assert rows.ndim == 2 and rows.shape[1] == 15 and len(rows) == len(prob)
    h = hashlib.sha256(); h.update(rows.tobytes()); h.update(prob.tobytes())
    change = np.flatnonzero((np.diff(rows[:, :5].astype(np.int16), axis=0) != 0).any(axis=1)) + 1
    bounds = np.concatenate([[0], change, [len(rows)]])
    per = np.zeros(len(bounds) - 1, np.uint32)
    for i in range(len(bounds) - 1):
        s, e = bounds[i], bounds[i + 1]
        per[i] = int.from_bytes(hashlib.sha256(rows[s:e].tobytes() + prob[s:e].tobytes()).digest()[:4], "little")
    return h.hexdigest(), per


def dump_table(Env, width, height, slip, digest_only=False):
    t0 = time.time()
    env = Env(width=width, height=height, slip_prob=slip)
    H, W = env.height, env.width
    n_tuples = H * W * H * W * 2
    lut = np.full(n_tuples, 0xFFFF, dtype=np.uint16)
    kind = np.zeros(n_tuples, dtype=np.uint8)       # 0 unreachable, 1 live, 2 goal
    goal_value = np.zeros(n_tuples, dtype=np.int8)

    def flat(st):
        xa, ya, xb, yb, p = st
        return (((xa * W + ya) * H + xb) * W + yb) * 2 + p

    for st, idx in env.state_space.items():
        if st == env.TERMINAL_STATE:
            continue
        lut[flat(st)] = idx
        kind[flat(st)] = 1
    for st, val in env.goal_states.items():
        lut[flat(st)] = 0
        kind[flat(st)] = 2
        goal_value[flat(st)] = int(val)
    for st in env.unreachable_states:
        assert kind[flat(st)] == 0

    rows, probs = [], []
    for st, per_action in env.P_readable.items():
        for (asa, asb), trans in per_action.items():
            aa = env.ACTION_STRING_TO_INT[asa]
            ab = env.ACTION_STRING_TO_INT[asb]
            for k, (p, ns, r, d) in enumerate(trans):
                rows.append(tuple(st) + (aa, ab, k) + tuple(ns) + (int(r), int(d)))
                probs.append(p)
    rows = np.asarray(rows, dtype=np.int8)
    probs = np.asarray(probs, dtype=np.float64)
    if digest_only:
        r = rows[rows[:, 7] == 0].astype(np.int64)
        key = ((((r[:, 0] * W + r[:, 1]) * H + r[:, 2]) * W + r[:, 3]) * 2 + r[:, 4]) * 25 + r[:, 5] * 5 + r[:, 6]
        assert (np.diff(key) > 0).all(), "the reference's iteration order is the canonical order"
        sha, per = table_digest(rows, probs)
        lens = np.diff(np.append(np.flatnonzero(rows[:, 7] == 0), len(rows)))
        out = os.path.join(HERE, "digest_%dx%d_s%s.npz" % (width, height, slip_tag(slip)))
        np.savez_compressed(
            out, width=np.int32(width), height=np.int32(height), slip=np.float64(slip),
            nS=np.int32(env.nS), nA=np.int32(env.nA), goal_rows=np.asarray(env.goal_rows, dtype=np.int8),
            lut=lut, kind=kind, goal_value=goal_value,
            isd_states=np.asarray([s for _, s in env.isd], dtype=np.int8),
            isd_probs=np.asarray([p for p, _ in env.isd], dtype=np.float64),
            n_rows=np.int64(len(rows)), sha256=np.bytes_(sha), tuple_digest=per,
            list_length_hist=np.bincount(lens, minlength=37).astype(np.int64),
            prob_sum=np.float64(probs.sum()))            # (no timings in here: regenerating must reproduce the file byte for byte)
        print("  %s: %d rows (not stored), sha256 %s, nS=%d, %.1fs, %d KB" % (
            os.path.basename(out), len(rows), sha[:16], env.nS, time.time() - t0, os.path.getsize(out) // 1024))
        return env
    out = os.path.join(HERE, "table_%dx%d_s%s.npz" % (width, height, slip_tag(slip)))
    np.savez_compressed(
        out,
        width=np.int32(width), height=np.int32(height), slip=np.float64(slip),
        nS=np.int32(env.nS), nA=np.int32(env.nA),
        goal_rows=np.asarray(env.goal_rows, dtype=np.int8),
        lut=lut, kind=kind, goal_value=goal_value,
        rows=rows, prob=probs,
        isd_states=np.asarray([s for _, s in env.isd], dtype=np.int8),
        isd_probs=np.asarray([p for p, _ in env.isd], dtype=np.float64),
    )
    print("  %s: %d rows, nS=%d, %.1fs, %d KB" % (
        os.path.basename(out), len(rows), env.nS, time.time() - t0, os.path.getsize(out) // 1024))
    return env


def dump_replay(env, width, height, slip, n, seed):
    """Random (state, t, joint action, u) through the real step()."""
    rng = np.random.RandomState(seed)
    keys = list(env.P_readable.keys())           # live + goal tuples
    live = [k for k in keys if k not in env.goal_states]
    st_in = np.zeros((n, 5), np.int8); t_in = np.zeros(n, np.int16)
    act = np.zeros((n, 2), np.int8); u_in = np.zeros(n, np.float64)
    st_out = np.zeros((n, 5), np.int8); obs = np.zeros(n, np.uint16)
    r_a = np.zeros(n, np.float64); r_b = np.zeros(n, np.float64)
    done = np.zeros(n, np.uint8); trunc = np.zeros(n, np.uint8)
    p_info = np.zeros(n, np.float64); needs_reset = np.zeros(n, np.uint8)
    special_u = [0.0, 0.25, 0.5, 0.75, 0.2499999999999999, 0.4999999999999999,
                 0.9999999999999999, 1.0 - 2.0 ** -53, 2.0 ** -53]
    if slip > 0:
        c0 = (1 - slip) * (1 - slip)
        c1 = (1 - slip) * slip * 0.5
        special_u += [c0, np.nextafter(c0, 0), c0 + c1, c0 + c1 + c1, 0.9999, 0.99999]
    for i in range(n):
        st = keys[rng.randint(len(keys))] if rng.rand() < 0.05 else live[rng.randint(len(live))]
        t = [rng.randint(0, 100), 98, 99, 0][rng.randint(4)] if rng.rand() < 0.3 else rng.randint(0, 100)
        aa, ab = rng.randint(0, 5), rng.randint(0, 5)
        u = special_u[rng.randint(len(special_u))] if rng.rand() < 0.15 else rng.random_sample()
        env.state = st; env.timestep = t; env.needs_reset = False
        env.np_random = _FixedU(u)
        o, r, d, tr, info = env.step({"player_a": aa, "player_b": ab})
        assert o["player_a"] == o["player_b"]
        st_in[i] = st; t_in[i] = t; act[i] = (aa, ab); u_in[i] = u
        st_out[i] = env.state; obs[i] = o["player_a"]
        r_a[i] = r["player_a"]; r_b[i] = r["player_b"]
        done[i] = d["player_a"]; trunc[i] = tr["player_a"]
        p_info[i] = info["player_a"]["p"]; needs_reset[i] = env.needs_reset
    out = os.path.join(HERE, "replay_%dx%d_s%s.npz" % (width, height, slip_tag(slip)))
    np.savez_compressed(out, width=np.int32(width), height=np.int32(height), slip=np.float64(slip),
                        state=st_in, t=t_in, action=act, u=u_in,
                        next_state=st_out, obs=obs, reward_a=r_a, reward_b=r_b,
                        terminated=done, truncated=trunc, p=p_info, needs_reset=needs_reset)
    print("  %s: %d vectors, %d KB" % (os.path.basename(out), n, os.path.getsize(out) // 1024))


def dump_reset(env, width, height, slip=None):
    us = [0.0, 0.2499, 0.25, 0.4999999, 0.5, 0.75, 0.99999, 1.0 - 2.0 ** -53] + \
        list(np.random.RandomState(5).random_sample(200))
    states = np.zeros((len(us), 5), np.int8); obs = np.zeros(len(us), np.uint16)
    p = np.zeros(len(us), np.float64)
    for i, u in enumerate(us):
        env.np_random = _FixedU(u)
        o, info = env.reset()
        states[i] = env.state; obs[i] = o["player_a"]; p[i] = info["player_a"]["p"]
        assert env.timestep == 0 and env.needs_reset is False
    out = os.path.join(HERE, "reset_%dx%d%s.npz" % (width, height, "" if slip is None else "_s" + slip_tag(slip)))
    np.savez_compressed(out, width=np.int32(width), height=np.int32(height), slip=np.float64(slip or 0.0),
                        u=np.asarray(us, np.float64), state=states, obs=obs, p=p)
    print("  %s: %d vectors" % (os.path.basename(out), len(us)))


def dump_traj(Env, slip, seed, n_steps, tag=""):
    """MT19937-driven episodes through the real reset()/step() (BASELINE config 1 shape)."""
    env = Env(width=5, height=4, slip_prob=slip)
    rec = _RecordingRS(env.np_random)
    env.np_random = rec
    actions = np.random.RandomState(123).randint(0, 5, size=(n_steps, 2)).astype(np.int8)
    obs = np.zeros(n_steps, np.uint16); r_a = np.zeros(n_steps); r_b = np.zeros(n_steps)
    done = np.zeros(n_steps, np.uint8); trunc = np.zeros(n_steps, np.uint8)
    p = np.zeros(n_steps); state = np.zeros((n_steps, 5), np.int8)
    reset_before = np.zeros(n_steps, np.uint8); reset_obs = np.zeros(n_steps, np.uint16)
    u_step = np.zeros(n_steps); u_reset = np.full(n_steps, -1.0)
    o, _ = env.reset(seed=seed)
    first_obs = o["player_a"]; u_first_reset = rec.drawn[-1]
    for k in range(n_steps):
        if env.needs_reset:
            o, _ = env.reset()
            reset_before[k] = 1; reset_obs[k] = o["player_a"]; u_reset[k] = rec.drawn[-1]
        o, r, d, tr, info = env.step({"player_a": int(actions[k, 0]), "player_b": int(actions[k, 1])})
        u_step[k] = rec.drawn[-1]
        obs[k] = o["player_a"]; r_a[k] = r["player_a"]; r_b[k] = r["player_b"]
        done[k] = d["player_a"]; trunc[k] = tr["player_a"]; p[k] = info["player_a"]["p"]
        state[k] = env.state
    out = os.path.join(HERE, "traj_5x4_s%s_seed%d%s.npz" % (slip_tag(slip), seed, tag))
    np.savez_compressed(out, slip=np.float64(slip), seed=np.int64(seed), actions=actions,
                        first_obs=np.uint16(first_obs), u_first_reset=np.float64(u_first_reset),
                        obs=obs, reward_a=r_a, reward_b=r_b, terminated=done, truncated=trunc,
                        p=p, state=state, reset_before=reset_before, reset_obs=reset_obs,
                        u_step=u_step, u_reset=u_reset)
    print("  %s: %d steps, %d episodes, %d KB" % (
        os.path.basename(out), n_steps, int(reset_before.sum()) + 1, os.path.getsize(out) // 1024))


def dump_single_agent(Env, slip, learner):
    """Single-agent mode: the other side follows a fixed dict policy baked into the table
    (soccer_simultaneous_env.py:54-56,187-188,243-244,266-279)."""
    from gym_soccer.utils.policies import get_random_policy
    policy = get_random_policy(761, 5, seed=0)
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    env = Env(width=5, height=4, slip_prob=slip, **kw)
    rows, probs = [], []
    for st, per_action in env.P_readable.items():
        for asx, trans in per_action.items():
            a = env.ACTION_STRING_TO_INT[asx]
            for k, (p, ns, r, d) in enumerate(trans):
                rows.append(tuple(st) + (a, k) + tuple(ns) + (int(r), int(d)))
                probs.append(p)
    out = os.path.join(HERE, "single_5x4_s%s_%s.npz" % (slip_tag(slip), learner))
    np.savez_compressed(out, slip=np.float64(slip), learner=np.bytes_(learner),
                        policy=np.asarray([policy[s] for s in range(761)], np.int8),
                        rows=np.asarray(rows, np.int8), prob=np.asarray(probs, np.float64))
    print("  %s: %d rows, %d KB" % (os.path.basename(out), len(rows), os.path.getsize(out) // 1024))


def main():
    _install_gym_stand_in()
    from gym_soccer.envs import SoccerSimultaneousEnv as Env
    print("reference imported from", REF)
    for (w, h, s, nrep) in [(5, 4, 0.0, 20000), (5, 4, 0.2, 20000), (5, 4, 0.5, 4000),
                            (5, 4, 1.0, 4000), (6, 4, 0.0, 4000), (7, 5, 0.0, 4000),
                            (7, 5, 0.3, 4000), (9, 6, 0.0, 4000)]:
        env = dump_table(Env, w, h, s)
        dump_replay(env, w, h, s, nrep, seed=1000 + w * 10 + h)
        if s == 0.0:
            dump_reset(env, w, h)
    for s in (0.0, 0.2):
        for seed in (0, 7):
            dump_traj(Env, s, seed, 3000)
    for learner in ("player_a", "player_b"):
        dump_single_agent(Env, 0.2, learner)




def dump_value_iteration(Env, slip, learner, opponent, tag):
    """The reference's own planner (gym_soccer/utils/planners.py:4-18) on the reference's own tables:
    synchronous value iteration of the learner's best response against a fixed opponent policy."""
    from gym_soccer.utils.planners import value_iteration
    from gym_soccer.utils.policies import get_random_policy, get_stand_policy
    policy = get_random_policy(761, 5, seed=0) if opponent == "random" else get_stand_policy(761)
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    env = Env(width=5, height=4, slip_prob=slip, **kw)
    t0 = time.time()
    pi, V, Q, cc = value_iteration(env, theta=1e-10, discount_factor=0.99)
    out = os.path.join(HERE, "vi_5x4_s%s_%s_vs_%s.npz" % (slip_tag(slip), learner, opponent))
    np.savez_compressed(out, slip=np.float64(slip), learner=np.bytes_(learner),
                        policy=np.asarray([policy[s] for s in range(761)], np.int8),
                        theta=np.float64(1e-10), discount_factor=np.float64(0.99),
                        pi=np.asarray(pi, np.int64), V=np.asarray(V, np.float64), Q=np.asarray(Q, np.float64),
                        iterations=np.int64(cc))
    print("  %s: %d iterations, %.0fs, %d KB" % (os.path.basename(out), cc, time.time() - t0, os.path.getsize(out) // 1024))
    return env, policy


def dump_other_planners(env, policy, slip, learner, opponent):
    """policy_evaluation / policy_improvement / policy_iteration / policy_eval / modified_policy_iteration
    (gym_soccer/utils/planners.py:20-87) of the reference on the same env; reference seconds are recorded for
    the speed comparison in DESIGN.md."""
    from gym_soccer.utils import planners as pl
    theta, gamma = 1e-10, 0.99
    rs = np.random.RandomState(11)
    pi_eval = rs.randint(0, 5, 761)
    sec = {}
    t0 = time.time(); V_eval = pl.policy_evaluation(pi_eval, env, theta, gamma); sec["policy_evaluation"] = time.time() - t0
    new_pi, Q_imp = pl.policy_improvement(V_eval, env, gamma)
    np.random.seed(0)
    pi0 = np.random.choice((0, 1, 2, 3, 4), 761)          # what policy_iteration draws first (:45)
    np.random.seed(0)
    t0 = time.time(); pi_pi, pi_V, pi_Q, pi_cc = pl.policy_iteration(env, theta, gamma); sec["policy_iteration"] = time.time() - t0
    stoch = rs.dirichlet(np.ones(5), 761)
    init = rs.standard_normal(761)
    pe_v, pe_cc = pl.policy_eval(env, stoch, theta, gamma, k=25, init=init.copy())
    pe0_v, pe0_cc = pl.policy_eval(env, stoch, 1e-6, 0.9)
    t0 = time.time(); m1 = pl.modified_policy_iteration(env, 1, theta, gamma); sec["mpi_k1"] = time.time() - t0
    t0 = time.time(); m2 = pl.modified_policy_iteration(env, 10000000, theta, gamma); sec["mpi_kinf"] = time.time() - t0
    m3 = pl.modified_policy_iteration(env, 5, 1e-6, 0.9)
    out = os.path.join(HERE, "planners_5x4_s%s_%s_vs_%s.npz" % (slip_tag(slip), learner, opponent))
    np.savez_compressed(
        out, slip=np.float64(slip), learner=np.bytes_(learner),
        policy=np.asarray([policy[s] for s in range(761)], np.int8), theta=np.float64(theta), discount_factor=np.float64(gamma),
        pe_pi=pi_eval.astype(np.int64), pe_V=V_eval, imp_pi=np.asarray(new_pi, np.int64), imp_Q=Q_imp,
        pi_pi0=pi0.astype(np.int64), pi_pi=np.asarray(pi_pi, np.int64), pi_V=pi_V, pi_Q=pi_Q, pi_iterations=np.int64(pi_cc),
        de_policy=stoch, de_init=init, de_v=pe_v, de_cc=np.int64(pe_cc), de0_v=pe0_v, de0_cc=np.int64(pe0_cc),
        mpi1_pi=np.asarray(m1[0], np.int64), mpi1_V=m1[1], mpi1_Q=m1[2], mpi1_counter=np.int64(m1[3]),
        mpi2_pi=np.asarray(m2[0], np.int64), mpi2_V=m2[1], mpi2_Q=m2[2], mpi2_counter=np.int64(m2[3]),
        mpi3_pi=np.asarray(m3[0], np.int64), mpi3_V=m3[1], mpi3_Q=m3[2], mpi3_counter=np.int64(m3[3]))
    # how long the reference took (for tools/planner_time.py): next to the fixture, not in it — the .npz must regenerate byte for byte
    with open(out[:-4] + "_reference_seconds.json", "w") as f:
        json.dump({k: round(v, 2) for k, v in sec.items()}, f)
    print("  %s: PI %d iterations, MPI counters %d / %d / %d, %s, %d KB"
          % (os.path.basename(out), pi_cc, m1[3], m2[3], m3[3], {k: round(v, 2) for k, v in sec.items()}, os.path.getsize(out) // 1024))


def main_planners():
    _install_gym_stand_in()
    from gym_soccer.envs import SoccerSimultaneousEnv as Env
    for slip, learner, opp in [(0.2, "player_a", "random"), (0.2, "player_b", "random"), (0.0, "player_a", "stand")]:
        env, policy = dump_value_iteration(Env, slip, learner, opp, "")
        if opp == "random":
            dump_other_planners(env, policy, slip, learner, opp)


def main_one(w, h, s, nrep):
    """One pitch only, e.g. `make_golden.py one 11 7 0.0 4000` — the reference's largest parametrisation
    (gym_soccer/tests/test_general.py:5-11); its constructor takes ~2 minutes."""
    _install_gym_stand_in()
    from gym_soccer.envs import SoccerSimultaneousEnv as Env
    env = dump_table(Env, w, h, s)
    dump_replay(env, w, h, s, nrep, seed=1000 + w * 10 + h)
    if s == 0.0:
        dump_reset(env, w, h)


def main_digest(w, h, s, nrep):
    """A pitch whose full table is too large to commit (11x7 with slip, the reference's largest parametrisation
    gym_soccer/tests/test_general.py:5-11 with the slip of its registration stub): digest + replay + reset."""
    _install_gym_stand_in()
    from gym_soccer.envs import SoccerSimultaneousEnv as Env
    env = dump_table(Env, w, h, s, digest_only=True)
    dump_replay(env, w, h, s, nrep, seed=1000 + w * 10 + h)
    dump_reset(env, w, h, slip=s)


def main_traj10k():
    """BASELINE config 1 at its stated length: 10 000 uniform-random joint-action steps, reset(seed=0)."""
    _install_gym_stand_in()
    from gym_soccer.envs import SoccerSimultaneousEnv as Env
    for s in (0.0, 0.2):
        dump_traj(Env, s, 0, 10000, tag="_10k")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "planners":
        main_planners()
    elif len(sys.argv) > 1 and sys.argv[1] == "digest":
        main_digest(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 4000)
    elif len(sys.argv) > 1 and sys.argv[1] == "traj10k":
        main_traj10k()
    elif len(sys.argv) > 1 and sys.argv[1] == "one":
        main_one(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 4000)
    else:
        main()
        main_planners()
