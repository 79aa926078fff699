"""Value iteration (reference gym_soccer/utils/planners.py:4-18): the oracle's restatement is pinned to
fixtures produced by the reference's own planner on its own env (tests/golden/make_golden.py planners),
and the device planner (soccer_value_iteration) must reproduce them bit for bit."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "vi_*.npz")))


def _load(path):
    d = np.load(path)
    return d, float(d["slip"]), bytes(d["learner"]).decode(), d["policy"]


def test_fixtures_present():
    assert len(FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_value_iteration_matches_reference(path):
    from oracle.oracle import Oracle, single_agent_lists, value_iteration
    d, slip, learner, policy = _load(path)
    orc = Oracle(5, 4, slip)
    P = single_agent_lists(orc, learner, policy)
    pi, V, Q, cc = value_iteration(P, orc.nS, float(d["theta"]), float(d["discount_factor"]))
    assert cc == int(d["iterations"])
    assert np.array_equal(V, d["V"]) and np.array_equal(Q, d["Q"])       # bit-exact float64
    assert np.array_equal(pi, d["pi"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_device_value_iteration_matches_reference(path):
    import gym_soccer_littman94_amd as gsa
    from gym_soccer_littman94_amd.planners import value_iteration
    d, slip, learner, policy = _load(path)
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    env = gsa.SoccerSimultaneousEnv(width=5, height=4, slip_prob=slip, **kw)
    pi, V, Q, cc = value_iteration(env, float(d["theta"]), float(d["discount_factor"]))
    assert cc == int(d["iterations"])
    assert np.array_equal(V, d["V"]) and np.array_equal(Q, d["Q"])       # bit-exact float64
    assert np.array_equal(pi, d["pi"])
    # and the planner agrees with the reference's planner run over the facade's own P tables
    # (the host loop the reference would execute), on a short horizon
    pi2, V2, Q2, cc2 = value_iteration(env, 1e-2, 0.9)
    from oracle.oracle import value_iteration as host_vi
    pi3, V3, Q3, cc3 = host_vi(env.P, env.nS, 1e-2, 0.9)
    assert cc2 == cc3 and np.array_equal(V2, V3) and np.array_equal(Q2, Q3) and np.array_equal(pi2, pi3)


@pytest.mark.gpu
def test_device_value_iteration_needs_single_agent_mode():
    import gym_soccer_littman94_amd as gsa
    from gym_soccer_littman94_amd.planners import value_iteration
    env = gsa.SoccerSimultaneousEnv()
    with pytest.raises(AssertionError):
        value_iteration(env, 1e-10, 0.99)
    with pytest.raises(AssertionError):
        env._batch.value_iteration(1e-10, 0.99)


@pytest.mark.gpu
def test_device_value_iteration_large_pitch_and_iteration_cap():
    import gym_soccer_littman94_amd as gsa
    from oracle.oracle import Oracle, single_agent_lists, value_iteration as host_vi
    rng = np.random.default_rng(5)
    b = gsa.SoccerBatch(1, 7, 5, 0.3)
    policy = rng.integers(0, 5, b.nS).astype(np.int8)
    b.set_policy("player_a", policy)
    pi, V, Q, cc = b.value_iteration(1e-10, 0.95, max_iterations=7)
    assert cc == 7
    orc = Oracle(7, 5, 0.3)
    pi2, V2, Q2, cc2 = host_vi(single_agent_lists(orc, "player_b", policy), orc.nS, 1e-10, 0.95, max_iterations=7)
    assert cc2 == 7 and np.array_equal(V, V2) and np.array_equal(Q, Q2) and np.array_equal(pi, pi2)
