"""-m gpu: the gym-style host layer (SoccerSimultaneousEnv facade, VectorSoccerEnv, make) on top of
the HIP path.  The known-answer vectors are the reference's own
(gym_soccer/tests/test_deterministic_soccer_simultaneous_env.py, restated as data — SURVEY.md
Appendix B); the API-shape checks follow gym_soccer/tests/test_general.py:159-302."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import gym_soccer_littman94_amd as gsa
from gym_soccer_littman94_amd import SoccerSimultaneousEnv, VectorSoccerEnv
from oracle.oracle import Oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJS = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
N, S, E, W, O = 1, 2, 3, 4, 0     # NORTH, SOUTH, EAST, WEST, NOOP


@pytest.fixture(scope="module")
def env():
    e = SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def slip_env():
    e = SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.2)
    yield e
    e.close()


def step_from(env, state, a, b):
    env.reset()
    env.state = state
    return env.step({'player_a': a, 'player_b': b})


# ---- drop-in proof: same seed, same actions => the reference's own MT19937 trajectories ----------
@pytest.mark.parametrize("path", TRAJS, ids=[os.path.basename(p)[:-4] for p in TRAJS])
def test_same_seed_reproduces_reference_trajectories(path):
    g = np.load(path)
    env = SoccerSimultaneousEnv(width=5, height=4, slip_prob=float(g["slip"]))
    obs, info = env.reset(seed=int(g["seed"]))
    assert obs['player_a'] == g["first_obs"] == obs['player_b'] and info['player_a']['p'] == 0.25
    n = len(g["obs"]) if path.endswith("_10k.npz") else 600       # _10k: BASELINE config 1 at its stated length, every step
    for k in range(n):
        if env.needs_reset:
            assert g["reset_before"][k]
            obs, _ = env.reset()
            assert obs['player_a'] == g["reset_obs"][k]
        else:
            assert not g["reset_before"][k]
        o, r, d, t, i = env.step({'player_a': int(g["actions"][k, 0]), 'player_b': int(g["actions"][k, 1])})
        assert o['player_a'] == o['player_b'] == g["obs"][k], k
        assert r['player_a'] == g["reward_a"][k] and r['player_b'] == g["reward_b"][k]
        assert d['player_a'] == bool(g["terminated"][k]) and t['player_b'] == bool(g["truncated"][k])
        assert i['player_a']['p'] == g["p"][k] == i['player_b']['p']
        assert env.state == tuple(g["state"][k])
    env.close()


def test_survey_appendix_b_seed7_vector(slip_env):
    """SURVEY.md Appendix B (measured on the reference, independent of tests/golden): slip_prob=0.2,
    reset(seed=7), then actions (k%5, 3k%5) for k = 0..9."""
    o, _ = slip_env.reset(seed=7)
    assert o['player_a'] == 253
    seen = []
    for k in range(10):
        o, r, d, t, i = slip_env.step({'player_a': k % 5, 'player_b': (3 * k) % 5})
        seen.append(o['player_a'])
    assert seen == [253, 65, 253, 453, 415, 415, 227, 407, 443, 403]


# ---- known-answer vectors of the reference's deterministic tests ---------------------------------
def test_initialization_and_shapes(env):
    assert env.width == 7 and env.height == 4 and env.slip_prob == 0.0
    assert env.action_space['player_a'].n == 5 and env.action_space['player_b'].n == 5
    assert env.observation_space['player_a'].n == 761 == env.nS
    assert len(env.state_space) == 761 and len(env.goal_states) == 160 and len(env.unreachable_states) == 648
    assert env.state_space[(0, 1, 0, 2, 0)] == 1 and env.state_space[(3, 5, 3, 4, 1)] == 760
    assert env.isd == [(0.25, (1, 2, 2, 4, 0)), (0.25, (1, 2, 2, 4, 1)), (0.25, (2, 2, 1, 4, 0)), (0.25, (2, 2, 1, 4, 1))]
    obs, info = env.reset()
    assert isinstance(obs, dict) and isinstance(info, dict) and set(obs) == set(info) == {'player_a', 'player_b'}
    assert obs['player_a'] in (253, 254, 435, 436)
    out = env.step({'player_a': O, 'player_b': O})
    assert len(out) == 5 and all(isinstance(x, dict) for x in out)


@pytest.mark.parametrize("state,a,b,ra", [
    ((1, 5, 3, 1, 0), E, O, +1), ((3, 5, 1, 1, 1), O, W, -1),                  # scoring (:49,:52)
    ((1, 1, 3, 5, 0), W, O, -1), ((2, 1, 3, 5, 0), W, O, -1),                  # own goals (:56-84)
    ((3, 1, 1, 5, 1), O, E, +1), ((3, 1, 2, 5, 1), O, E, +1),
    ((1, 5, 1, 1, 0), E, W, +1), ((1, 5, 1, 1, 1), E, W, -1),                  # simultaneous attempts (:343-352)
    ((1, 5, 3, 3, 0), E, O, +1), ((2, 1, 3, 3, 0), W, O, -1),                  # goal-edge scoring (:411-421)
])
def test_goals(env, state, a, b, ra):
    o, r, d, t, i = step_from(env, state, a, b)
    assert d['player_a'] and d['player_b']
    assert r['player_a'] == ra and r['player_b'] == -ra
    assert o['player_a'] == 0 and env.needs_reset
    assert isinstance(r['player_a'], float) and isinstance(d['player_a'], bool) and isinstance(t['player_a'], bool)


@pytest.mark.parametrize("state,a,b", [
    ((1, 2, 1, 3, 0), E, W), ((1, 2, 1, 3, 1), E, W),          # head-on swap (:89-100)
    ((1, 2, 1, 3, 0), E, O), ((1, 2, 1, 3, 1), O, W),          # into a stander (:105-116)
])
def test_collisions_leave_positions(env, state, a, b):
    step_from(env, state, a, b)
    assert env.state[:4] == state[:4] and env.state[4] in (0, 1)


def test_stander_collision_flips_possession(env):
    step_from(env, (1, 2, 1, 3, 0), E, O); assert env.state == (1, 2, 1, 3, 1)
    step_from(env, (1, 2, 1, 3, 1), O, W); assert env.state == (1, 2, 1, 3, 0)


@pytest.mark.parametrize("p", [0, 1])
@pytest.mark.parametrize("a,b,sa,sb", [(N, E, (0, 1), (3, 5)), (W, E, (0, 1), (3, 5)), (N, S, (0, 1), (3, 5)),
                                       (W, S, (0, 1), (3, 5)), (E, N, (3, 5), (0, 1)), (E, W, (3, 5), (0, 1)),
                                       (S, N, (3, 5), (0, 1)), (S, W, (3, 5), (0, 1))])
def test_walls_and_corners_bounce(env, p, a, b, sa, sb):      # :170-264
    state = sa + sb + (p,)
    step_from(env, state, a, b)
    assert env.state == state


@pytest.mark.parametrize("state,a,b", [
    ((1, 1, 3, 3, 1), W, O), ((2, 1, 3, 3, 1), W, O), ((3, 3, 1, 5, 0), O, E), ((3, 3, 2, 5, 0), O, E),
    ((3, 3, 1, 1, 0), O, W), ((3, 3, 2, 1, 0), O, W), ((1, 5, 3, 3, 1), E, O), ((2, 5, 3, 3, 1), E, O)])
def test_cannot_enter_goal_without_ball(env, state, a, b):    # :268-321
    o, r, d, t, i = step_from(env, state, a, b)
    assert env.state == state and not d['player_a'] and r['player_a'] == 0


def test_follow_and_out_of_bounds(env):
    for p in (0, 1):                                          # no possession change without collision (:333-339)
        step_from(env, (1, 1, 3, 3, p), E, W); assert env.state[4] == p
        step_from(env, (1, 1, 1, 2, p), E, E); assert env.state == (1, 2, 1, 3, p)   # follow the leader (:356-371)
        step_from(env, (1, 1, 1, 3, p), E, E); assert env.state == (1, 2, 1, 4, p)
    step_from(env, (0, 1, 3, 5, 0), N, E); assert env.state == (0, 1, 3, 5, 0)       # :398-407
    step_from(env, (0, 1, 3, 4, 1), N, E); assert env.state == (0, 1, 3, 5, 1)


@pytest.mark.parametrize("state,a,b", [((1, 1, 2, 2, 0), E, N), ((1, 2, 2, 1, 1), W, N), ((2, 1, 1, 2, 0), E, S),
                                       ((1, 1, 1, 3, 1), E, W), ((1, 3, 1, 1, 0), W, E), ((1, 1, 3, 1, 0), S, N),
                                       ((3, 1, 1, 1, 1), N, S)])
def test_same_target_cell_four_way_tie(env, state, a, b):     # :146-165, frequencies 0.45..0.55 over 1000 trials
    moved_a = moved_b = switched = 0
    for _ in range(1000):
        step_from(env, state, a, b)
        moved_a += env.state[:2] != state[:2]; moved_b += env.state[2:4] != state[2:4]
        switched += env.state[4] != state[4]
    assert moved_a + moved_b == 1000
    for c in (moved_a, moved_b, switched):
        assert 450 <= c <= 550


def test_captured_vectors_with_injected_uniforms(env):
    """SURVEY.md Appendix B 'additional vectors' (measured on the reference with an injected u)."""
    class U:
        def __init__(self, u): self.u = u
        def random(self): return self.u
        def seed(self, s): pass
    rs = env.np_random
    try:
        for u, exp, ob in [(0.49999, (1, 2, 1, 3, 0), 241), (0.5, (1, 2, 1, 3, 1), 242)]:
            env.np_random = rs; env.reset(); env.state = (1, 2, 1, 3, 0); env.np_random = U(u)
            o, r, d, t, i = env.step({'player_a': E, 'player_b': W})
            assert env.state == exp and o['player_a'] == ob and i['player_a']['p'] == 0.5
        for u, exp, ob in [(0.0, (1, 1, 1, 2, 0), 201), (0.25, (1, 1, 1, 2, 1), 202),
                           (0.5, (1, 2, 2, 2, 0), 249), (0.75, (1, 2, 2, 2, 1), 250)]:
            env.np_random = rs; env.reset(); env.state = (1, 1, 2, 2, 0); env.np_random = U(u)
            o, r, d, t, i = env.step({'player_a': E, 'player_b': N})
            assert env.state == exp and o['player_a'] == ob and i['player_a']['p'] == 0.25
        env.np_random = rs; env.reset(); env.state = (1, 2, 2, 4, 0); env.timestep = 99
        o, r, d, t, i = env.step({'player_a': O, 'player_b': O})
        assert t['player_a'] and not d['player_a'] and env.needs_reset
        env.reset(); env.state = (1, 6, 3, 3, 0)                 # stepping from a goal tuple: absorbing
        o, r, d, t, i = env.step({'player_a': E, 'player_b': O})
        assert o['player_a'] == 0 and r['player_a'] == 0 and d['player_a'] and env.state == (1, 6, 3, 3, 0)
        for u, ob in [(0.0, 253), (0.2499, 253), (0.25, 254), (0.5, 435), (0.75, 436), (0.99999, 436)]:
            env.np_random = U(u)
            assert env.reset()[0]['player_a'] == ob
    finally:
        env.np_random = rs


def test_assertions_like_the_reference(env):
    fresh = SoccerSimultaneousEnv()
    with pytest.raises(AssertionError, match="reset"):
        fresh.step({'player_a': 0, 'player_b': 0})             # :376
    fresh.close()
    env.reset()
    for bad in ([0, 0], {'player_a': 0}, {'player_a': 0, 'x': 1}, {'player_a': 0, 'player_b': 0, 'c': 0}):
        with pytest.raises(AssertionError):
            env.step(bad)
    step_from(env, (1, 5, 3, 1, 0), E, O)
    with pytest.raises(AssertionError, match="reset"):          # done -> needs reset (:406)
        env.step({'player_a': 0, 'player_b': 0})
    env.reset(); env.state = (1, 1, 1, 1, 0)
    with pytest.raises(KeyError):                                # no table entry for unreachable tuples (:394)
        env.step({'player_a': 0, 'player_b': 0})
    for kw in (dict(width=4), dict(height=3), dict(player_a_policy={0: 0}, player_b_policy={0: 0})):
        with pytest.raises(AssertionError):
            SoccerSimultaneousEnv(**kw)


@pytest.mark.parametrize("learner", ["player_a", "player_b"])
def test_single_agent_mode_matches_reference_tables(learner):
    """Fixed-opponent mode (reference :54-56, :187-188, :243-244): the sampled transition must be an
    entry of the reference's single-agent table for (state, learner action), with the reward sign of
    that table, and over many draws every entry must show up."""
    g = np.load(os.path.join(GOLDEN, "single_5x4_s0p2_%s.npz" % learner))
    policy = {s: int(a) for s, a in enumerate(g["policy"])}
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    env = SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.2, **kw)
    assert not env.multiagent and env.return_agent == [learner]
    assert list(env.observation_space) == [learner] and env.action_space[learner].n == 5
    rows = g["rows"]       # xa,ya,xb,yb,p, a, k, nxa,nya,nxb,nyb,np, reward, done
    table = {}
    for row in rows:
        table.setdefault((tuple(row[:5]), int(row[5])), set()).add((tuple(row[7:12]), int(row[12]), int(row[13])))
    rng = np.random.RandomState(0)
    keys = [k for k in table if k[0] not in env.goal_states]
    for idx in rng.choice(len(keys), 300, replace=False):
        st, a = keys[idx]
        env.reset(); env.state = st
        o, r, d, t, i = env.step({learner: a})
        assert set(o) == set(r) == set(d) == set(t) == set(i) == {learner}
        assert (env.state, int(r[learner]), int(d[learner])) in table[(st, a)]
    env.close()


TABLES = sorted(glob.glob(os.path.join(GOLDEN, "table_*.npz")))


@pytest.mark.parametrize("path", TABLES, ids=[os.path.basename(p)[:-4] for p in TABLES])
def test_device_enumerated_transition_table_equals_reference_row_for_row(path):
    """soccer_enumerate_transitions (the kernels' own rule functions, run over every state tuple x joint
    action) against the reference's complete P_readable: list order, float64 probabilities, next tuple,
    reward, done."""
    from gym_soccer_littman94_amd import SoccerBatch
    g = np.load(path)
    b = SoccerBatch(1, int(g["width"]), int(g["height"]), float(g["slip"]))
    count, prob, nxt, rew, done = b.transitions()
    W = int(g["width"]) + 2; H = int(g["height"])
    rows, gp = g["rows"], g["prob"]          # xa,ya,xb,yb,p, aa,ab, k, nxa,nya,nxb,nyb,np, reward, done
    r64 = rows.astype(np.int64)
    flat = (((r64[:, 0] * W + r64[:, 1]) * H + r64[:, 2]) * W + r64[:, 3]) * 2 + r64[:, 4]
    nflat = (((r64[:, 8] * W + r64[:, 9]) * H + r64[:, 10]) * W + r64[:, 11]) * 2 + r64[:, 12]
    ja = r64[:, 5] * 5 + r64[:, 6]; k = r64[:, 7]
    np.testing.assert_array_equal(prob[flat, ja, k], gp)                      # bit-exact float64
    np.testing.assert_array_equal(nxt[flat, ja, k], nflat)
    np.testing.assert_array_equal(rew[flat, ja, k], rows[:, 13])
    np.testing.assert_array_equal(done[flat, ja, k], rows[:, 14])
    # list lengths: total entries and per-key counts agree, unreachable tuples have no key
    assert int(count[count > 0].sum()) == len(rows)
    last = np.append(k[1:] == 0, True)
    np.testing.assert_array_equal(count[flat[last], ja[last]], k[last] + 1)
    np.testing.assert_array_equal(count[:, 0] < 0, g["kind"] == 0)
    b.close()


DIGESTS = sorted(glob.glob(os.path.join(GOLDEN, "digest_*.npz")))


@pytest.mark.parametrize("path", DIGESTS, ids=[os.path.basename(p)[:-4] for p in DIGESTS])
def test_device_enumerated_transition_table_hashes_to_the_reference_digest(path):
    """Tables too large to commit (11x7 with slip_prob 0.2: 3.4 M rows): soccer_enumerate_transitions, brought into the
    reference's own iteration order, must hash to what the reference's P_readable hashed to
    (tests/golden/make_golden.py::table_digest; soccer_simultaneous_env.py:167-293, sizes of tests/test_general.py:5-11)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import table_digest
    from gym_soccer_littman94_amd import SoccerBatch
    d = np.load(path)
    w, h = int(d["width"]), int(d["height"]); W = w + 2
    b = SoccerBatch(1, w, h, float(d["slip"]))
    count, prob, nxt, rew, done = b.transitions()
    lut, goal_value, isd = b.tables()
    np.testing.assert_array_equal(lut, d["lut"]); np.testing.assert_array_equal(goal_value, d["goal_value"])
    np.testing.assert_array_equal(isd, d["isd_states"]); assert b.nS == int(d["nS"])
    np.testing.assert_array_equal(count[:, 0] < 0, d["kind"] == 0)
    f, ja = np.nonzero(count > 0)                                      # ascending (flat tuple, joint action): the canonical order
    c = count[f, ja]
    assert int(c.sum()) == int(d["n_rows"])
    np.testing.assert_array_equal(np.bincount(c, minlength=37), d["list_length_hist"])
    F = np.repeat(f, c); JA = np.repeat(ja, c)
    K = np.arange(len(F)) - np.repeat(np.cumsum(c) - c, c)

    def unflat(x):
        p = x & 1; x = x >> 1; yb = x % W; x //= W; xb = x % h; x //= h; ya = x % W; xa = x // W
        return [xa, ya, xb, yb, p]
    cols = unflat(F) + [JA // 5, JA % 5, K] + unflat(nxt[F, JA, K].astype(np.int64)) + [rew[F, JA, K], done[F, JA, K]]
    rows = np.stack([np.asarray(x).astype(np.int8) for x in cols], axis=1)
    sha, per = table_digest(rows, prob[F, JA, K])
    bad = np.flatnonzero(per != d["tuple_digest"])
    assert bad.size == 0, "first differing state tuple (in order of appearance): %d" % bad[0]
    assert sha == d["sha256"].item().decode()
    b.close()


def test_facade_tables_have_the_reference_shape_and_quirks(slip_env):
    g = np.load(os.path.join(GOLDEN, "table_5x4_s0p2.npz"))
    env = slip_env
    PR, P, Pmat, Rmat = env.P_readable, env.P, env.Pmat, env.Rmat
    assert len(PR) == 920 and len(P) == 761 and Pmat.shape == (761, 761, 5, 5) and Rmat.shape == (761, 5, 5)
    tr = PR[(1, 2, 1, 3, 0)][('EAST', 'WEST')]
    assert isinstance(tr[0][0], float) and isinstance(tr[0][1], tuple) and isinstance(tr[0][3], bool)
    assert [t[0] for t in tr][:2] == [0.32000000000000006, 0.32000000000000006]
    assert P[241][(3, 4)][0][1] in (241, 242)
    assert abs(Pmat[0, 0, 0, 0] - 160.0) < 1e-9             # one unit per goal tuple
    np.testing.assert_allclose(Pmat[1:].sum(axis=1), 1.0, atol=1e-12)
    assert abs(Rmat).max() <= 1.0 and Rmat[0].max() == 0
    # spot-check against the golden rows
    rows, gp = g["rows"], g["prob"]
    for i in np.random.RandomState(0).choice(len(rows), 500, replace=False):
        st = tuple(int(x) for x in rows[i, :5]); key = (env.ACTION_STRING[rows[i, 5]], env.ACTION_STRING[rows[i, 6]])
        p_, ns_, r_, d_ = PR[st][key][int(rows[i, 7])]
        assert p_ == gp[i] and ns_ == tuple(int(x) for x in rows[i, 8:13]) and r_ == rows[i, 13] and d_ == bool(rows[i, 14])


def test_reference_style_planner_runs_on_our_tables():
    """Value iteration in the reference's formulation (gym_soccer/utils/planners.py:4-18 reads env.P only)
    against a stand-still opponent: the best response must score every episode (tests/test_general.py:304+)."""
    stand = {s: 0 for s in range(761)}
    env = SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.0, player_b_policy=stand)
    P = env.P
    V = np.zeros(len(P)); gamma = 0.99
    for _ in range(200):
        Q = np.zeros((len(P), 5))
        for s_ in range(len(P)):
            for a in range(5):
                for pr, ns, r, d in P[s_][a]:
                    Q[s_, a] += pr * (r + gamma * V[ns] * (not d))
        if np.max(np.abs(V - Q.max(1))) < 1e-10:
            break
        V = Q.max(1)
    pi = Q.argmax(1)
    wins = 0
    for ep in range(50):
        o, _ = env.reset()
        for _ in range(100):
            o, r, d, t, _ = env.step({'player_a': int(pi[o['player_a']])})
            if d['player_a'] or t['player_a']:
                wins += r['player_a'] == 1.0
                break
    assert wins == 50
    env.close()


def test_slip_probabilities_show_up_in_info_p(slip_env):
    seen = set()
    for _ in range(400):
        slip_env.reset(); slip_env.state = (1, 2, 2, 4, 0)
        o, r, d, t, i = slip_env.step({'player_a': E, 'player_b': W})
        seen.add(float(i['player_a']['p']))
    assert seen <= {0.64, 0.08, 0.01} and 0.64 in seen and 0.08 in seen
    # NOOP never slips (tests/test_slip_soccer_simultaneous_env.py:198-210)
    for _ in range(50):
        slip_env.reset(); slip_env.state = (1, 2, 3, 4, 0)
        slip_env.step({'player_a': O, 'player_b': O})
        assert slip_env.state == (1, 2, 3, 4, 0)


def test_render_smoke(env, capsys):
    env.reset(); env.state = (1, 5, 3, 1, 0)
    env.step({'player_a': E, 'player_b': O})
    env.render()
    out = capsys.readouterr().out
    assert "GOAL! Player A scored!" in out and "Ball possession: A" in out


# ---- VectorSoccerEnv ------------------------------------------------------------------------------
def test_vector_env_numpy_matches_oracle_and_gym_autoreset_convention():
    n, T = 2048, 150
    rng = np.random.default_rng(1)
    v = VectorSoccerEnv(n, slip_prob=0.2, seed=11)
    o = Oracle(5, 4, 0.2, n=n, seed=11, autoreset=True)
    assert v.num_envs == n and v.single_observation_space['player_a'].n == 761
    assert v.action_space['player_b'].nvec.shape == (n,)
    with pytest.raises(AssertionError, match="reset"):
        v.step({'player_a': np.zeros(n, int), 'player_b': np.zeros(n, int)})
    obs, info = v.reset(seed=11)
    np.testing.assert_array_equal(obs['player_a'], o.reset())
    assert info['player_b']['p'].shape == (n,) and (info['player_a']['p'] == 0.25).all()
    saw_final = 0
    for k in range(T):
        a = rng.integers(0, 5, size=(2, n))
        ob, rw, te, tr, inf = v.step({'player_a': a[0], 'player_b': a[1]})
        c = o.step(a[0], a[1])
        np.testing.assert_array_equal(ob['player_a'], c["obs"]); assert ob['player_b'] is ob['player_a']
        np.testing.assert_array_equal(rw['player_a'], c["reward"].astype(np.float32))
        np.testing.assert_array_equal(rw['player_b'], -c["reward"].astype(np.float32))
        assert rw['player_a'].dtype == np.float32 and te['player_a'].dtype == np.bool_ and tr['player_a'].dtype == np.bool_
        np.testing.assert_array_equal(te['player_a'], c["terminated"].astype(bool))
        np.testing.assert_array_equal(tr['player_b'], c["truncated"].astype(bool))
        np.testing.assert_array_equal(inf['player_a']['p'], np.round(c["prob"], 2))
        fin = inf["_final_observation"]
        np.testing.assert_array_equal(fin, (c["terminated"] | c["truncated"]).astype(bool))
        np.testing.assert_array_equal(inf["final_observation"]['player_a'], c["final_obs"])
        # finished lanes already show the first observation of their next episode
        assert np.isin(ob['player_a'][fin], (253, 254, 435, 436)).all()
        assert (inf["final_observation"]['player_a'][te['player_a']] == 0).all()
        saw_final += int(fin.sum())
    assert saw_final > n and int(v.episode_histogram().sum()) == saw_final
    for bad in ({'player_a': a[0]}, [a[0], a[1]], {'player_a': a[0] + 5, 'player_b': a[1]},
                {'player_a': a[0][:-1], 'player_b': a[1]}):
        with pytest.raises(AssertionError):
            v.step(bad)
    v.close()


@pytest.mark.parametrize("learner", ["player_a", "player_b"])
def test_vector_env_single_agent_mode(learner):
    n, T = 1024, 80
    rng = np.random.default_rng(31)
    policy = {s: int(a) for s, a in enumerate(rng.integers(0, 5, size=761))}
    parr = np.array([policy[s] for s in range(761)], np.int8)
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    v = VectorSoccerEnv(n, slip_prob=0.2, seed=2, **kw)
    o = Oracle(5, 4, 0.2, n=n, seed=2, autoreset=True)
    assert not v.multiagent and v.return_agent == [learner] and list(v.action_space) == [learner]
    obs, info = v.reset(seed=2)
    cur = o.reset()
    assert set(obs) == set(info) == {learner}
    np.testing.assert_array_equal(obs[learner], cur)
    for k in range(T):
        a = rng.integers(0, 5, size=n)
        ob, rw, te, tr, inf = v.step({learner: a})
        c = o.step(a, parr[cur]) if learner == "player_a" else o.step(parr[cur], a)
        assert set(ob) == set(rw) == set(te) == set(tr) == {learner}
        np.testing.assert_array_equal(ob[learner], c["obs"])
        sign = 1.0 if learner == "player_a" else -1.0
        np.testing.assert_array_equal(rw[learner], sign * c["reward"].astype(np.float32))
        np.testing.assert_array_equal(te[learner], c["terminated"].astype(bool))
        cur = c["obs"]
    with pytest.raises(AssertionError):
        v.step({'player_a': a, 'player_b': a})
    with pytest.raises(AssertionError, match="Both players"):
        VectorSoccerEnv(8, player_a_policy=policy, player_b_policy=policy)
    v.close()


def test_vector_env_without_autoreset_is_strict_like_the_reference():
    n = 512
    rng = np.random.default_rng(2)
    v = VectorSoccerEnv(n, slip_prob=0.0, seed=3, autoreset=False)
    v.reset()
    done = np.zeros(n, bool)
    with pytest.raises(AssertionError, match="reset"):
        for k in range(120):
            a = rng.integers(0, 5, size=(2, n))
            ob, rw, te, tr, inf = v.step({'player_a': a[0], 'player_b': a[1]})
            done |= te['player_a'] | tr['player_a']
    assert done.any()
    s = v.get_state()
    need = s["needs_reset"].astype(bool)
    assert (need | ~done).all()          # every lane we saw finish is parked (the last step may add more)
    ob, _ = v.reset(mask=need)
    assert not v.get_state()["needs_reset"].any()
    v.close()


@pytest.mark.parametrize("slip", [0.2, 0.0])
def test_vector_env_device_io_matches_numpy_io(slip):
    import torch
    n, T = 4096, 60
    rng = np.random.default_rng(4)
    vn = VectorSoccerEnv(n, slip_prob=slip, seed=21)
    vd = VectorSoccerEnv(n, slip_prob=slip, seed=21, io="device")
    on, _ = vn.reset(); od, _ = vd.reset()
    np.testing.assert_array_equal(on['player_a'], od['player_a'].cpu().numpy().astype(np.uint16))
    for k in range(T):
        a = rng.integers(0, 5, size=(2, n)).astype(np.int8)
        rn = vn.step({'player_a': a[0], 'player_b': a[1]})
        ta = torch.from_numpy(a[0]).cuda(); tb = torch.from_numpy(a[1]).cuda()
        rd = vd.step({'player_a': ta, 'player_b': tb})
        np.testing.assert_array_equal(rn[0]['player_a'], rd[0]['player_a'].cpu().numpy().astype(np.uint16))
        np.testing.assert_array_equal(rn[1]['player_b'], rd[1]['player_b'].cpu().numpy())
        np.testing.assert_array_equal(rn[2]['player_a'], rd[2]['player_a'].cpu().numpy())
        np.testing.assert_array_equal(rn[3]['player_a'], rd[3]['player_a'].cpu().numpy())
        np.testing.assert_array_equal(rn[4]['player_a']['p'], rd[4]['player_a']['p'].cpu().numpy())
        np.testing.assert_array_equal(rn[4]["_final_observation"], rd[4]["_final_observation"].cpu().numpy())
        np.testing.assert_array_equal(rn[4]["final_observation"]['player_b'],
                                      rd[4]["final_observation"]['player_b'].cpu().numpy().astype(np.uint16))
        np.testing.assert_array_equal(rn[1]['player_a'], rd[1]['player_a'].cpu().numpy())
        assert rd[1]['player_a'].dtype == torch.float32 and rd[2]['player_b'].dtype == torch.bool
        assert set(rd[4].keys()) == {'player_a', 'player_b', 'final_observation', '_final_observation'}
        np.testing.assert_array_equal(vd.reward_int8.cpu().numpy(), rn[1]['player_a'].astype(np.int8))
    np.testing.assert_array_equal(vn.episode_histogram(), vd.episode_histogram())
    assert vn.episode_histogram().sum() > 0
    vn.close(); vd.close()


def test_checkpoint_resume_reproduces_the_future():
    n = 4096
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 5, size=(90, 2, n))
    v = VectorSoccerEnv(n, slip_prob=0.2, seed=5)
    v.reset()
    for k in range(40):
        v.step({'player_a': acts[k, 0], 'player_b': acts[k, 1]})
    ck = v.checkpoint()
    fut = [v.step({'player_a': acts[k, 0], 'player_b': acts[k, 1]}) for k in range(40, 90)]
    w = VectorSoccerEnv(n, slip_prob=0.2, seed=123)           # a different env, different seed
    # a checkpoint of another RNG convention (or of unknown convention) is refused: same (seed, tick), different random stream
    for bad in (dict(ck, rng_abi=2), {k: x for k, x in ck.items() if k != "rng_abi"}):
        with pytest.raises(AssertionError, match="RNG ABI"):
            w.restore(bad)
    w.restore(ck)
    for k, ref in zip(range(40, 90), fut):
        got = w.step({'player_a': acts[k, 0], 'player_b': acts[k, 1]})
        np.testing.assert_array_equal(got[0]['player_a'], ref[0]['player_a'])
        np.testing.assert_array_equal(got[1]['player_a'], ref[1]['player_a'])
        np.testing.assert_array_equal(got[2]['player_a'], ref[2]['player_a'])
        np.testing.assert_array_equal(got[3]['player_a'], ref[3]['player_a'])
    v.close(); w.close()


def test_make_ids():
    e = gsa.make("SoccerLittman94-v0")
    assert isinstance(e, SoccerSimultaneousEnv) and e.slip_prob == 0.0
    e.close()
    e = gsa.make("SoccerSimultaneous-v0")
    assert e.slip_prob == 0.2 and e.width == 7
    e.close()
    v = gsa.make("VectorSoccerLittman94-v0", num_envs=64)
    assert isinstance(v, VectorSoccerEnv) and v.num_envs == 64
    v.close()
    with pytest.raises(KeyError):
        gsa.make("WalkFive-v0")


def test_vector_env_has_the_rest_of_gyms_vector_surface():
    """step_async / step_wait, reset_async / reset_wait, is_vector_env, unwrapped, closed, the context manager: what wrappers
    written against gym 0.26's gym.vector.VectorEnv look for (the reference itself has no vector env)."""
    n = 256
    rng = np.random.default_rng(2)
    with VectorSoccerEnv(n, slip_prob=0.2, seed=4) as v, VectorSoccerEnv(n, slip_prob=0.2, seed=4) as w:
        assert v.is_vector_env and v.unwrapped is v and not v.closed and v.render_mode is None and "num_envs=256" in repr(v)
        v.reset_async(seed=9); o1, _ = v.reset_wait()
        o2, _ = w.reset(seed=9)
        np.testing.assert_array_equal(o1["player_a"], o2["player_a"])
        for _ in range(5):
            a = {"player_a": rng.integers(0, 5, n), "player_b": rng.integers(0, 5, n)}
            v.step_async(a); r1 = v.step_wait()
            r2 = w.step(a)
            for x, y in zip(r1[:4], r2[:4]):
                np.testing.assert_array_equal(x["player_a"], y["player_a"])
        with pytest.raises(AssertionError, match="without step_async"):
            v.step_wait()
    assert v.closed and w.closed
