"""The reference's own acceptance tests (gym_soccer/tests/test_general.py, test_slip_soccer_simultaneous_env.py),
restated as data and run against this package: the statistical slip scenarios with the reference's tolerance
bands (100 000 lanes per scenario in one batched step instead of 100 000 Python iterations), the initial-state
checks on the five pitch sizes, the structure of env.P, the single-/multi-agent surface, and value iteration
against a standing / random opponent followed by 1 000 played episodes (the planner runs on the device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NOOP, NORTH, SOUTH, EAST, WEST = 0, 1, 2, 3, 4
PITCHES = [(5, 4), (6, 4), (7, 5), (9, 6), (11, 7)]                     # test_general.py:5-11


def _one_step_batch(state, a, b, slip=0.2, n=100000, seed=0):
    """n lanes all in `state`, one step with the joint action: returns (next tuples [n, 5], terminated [n])."""
    from gym_soccer_littman94_amd import SoccerBatch
    bt = SoccerBatch(n, 5, 4, slip, seed=seed)
    bt.set_state(*[np.full(n, v, np.int8) for v in state[:4]], np.full(n, state[4], np.uint8), t=0, needs_reset=0)
    out = bt.step_host(np.full(n, a, np.int8), np.full(n, b, np.int8))
    s = bt.get_state(); bt.close()
    return np.stack([s["row_a"], s["col_a"], s["row_b"], s["col_b"], s["poss"]], 1), out["terminated"], out["reward"]


# ---- test_slip_soccer_simultaneous_env.py ----------------------------------------------------------------
@pytest.mark.parametrize("state,a,b", [((1, 5, 3, 1, 0), EAST, NOOP), ((3, 5, 1, 1, 1), NOOP, WEST)])   # :39-59
def test_scoring_ratio_under_slip(state, a, b):
    nxt, term, rew = _one_step_batch(state, a, b, seed=1)
    assert (np.abs(rew[term != 0]) == 1).all()
    assert 0.75 <= term.mean() <= 0.85


SLIP_GOAL = [((1, 1, 3, 3, 0), NORTH, NOOP), ((2, 1, 3, 3, 0), NORTH, NOOP), ((1, 1, 3, 3, 0), SOUTH, NOOP), ((2, 1, 3, 3, 0), SOUTH, NOOP),
             ((1, 5, 3, 3, 0), NORTH, NOOP), ((2, 5, 3, 3, 0), NORTH, NOOP), ((1, 5, 3, 3, 0), SOUTH, NOOP), ((2, 5, 3, 3, 0), SOUTH, NOOP),
             ((3, 3, 1, 1, 1), NOOP, NORTH), ((3, 3, 2, 1, 1), NOOP, NORTH), ((3, 3, 1, 1, 1), NOOP, SOUTH), ((3, 3, 2, 1, 1), NOOP, SOUTH),
             ((3, 3, 1, 5, 1), NOOP, NORTH), ((3, 3, 2, 5, 1), NOOP, NORTH), ((3, 3, 1, 5, 1), NOOP, SOUTH), ((3, 3, 2, 5, 1), NOOP, SOUTH)]


@pytest.mark.parametrize("k", range(len(SLIP_GOAL)))
def test_slip_into_goal(k):                                              # :83-119
    state, a, b = SLIP_GOAL[k]
    _, term, _ = _one_step_batch(state, a, b, seed=2 + k)
    assert 0.09 <= term.mean() <= 0.11


@pytest.mark.parametrize("state,a,b", [((0, 2, 3, 3, 0), NORTH, NOOP), ((0, 3, 3, 3, 0), NORTH, NOOP), ((3, 3, 0, 2, 1), NOOP, NORTH),
                                       ((3, 3, 0, 3, 1), NOOP, NORTH), ((3, 2, 0, 3, 0), SOUTH, NOOP), ((3, 3, 0, 3, 0), SOUTH, NOOP),
                                       ((0, 3, 3, 2, 0), NOOP, SOUTH), ((0, 3, 3, 3, 0), NOOP, SOUTH)])
def test_bounce_off_horizontal_edges(state, a, b):                       # :121-149
    nxt, _, _ = _one_step_batch(state, a, b, seed=30)
    stay = (nxt == np.array(state)).all(1).mean()
    assert 0.79 <= stay <= 0.81 and 0.19 <= 1 - stay <= 0.21


@pytest.mark.parametrize("state,a", [((0, 1, 3, 3, 1), WEST), ((3, 5, 0, 3, 1), EAST)])
def test_bounce_off_corner_edges(state, a):                              # :151-173
    nxt, _, _ = _one_step_batch(state, a, NOOP, seed=31)
    stay = (nxt == np.array(state)).all(1).mean()
    assert 0.89 <= stay <= 0.91 and 0.09 <= 1 - stay <= 0.11


@pytest.mark.parametrize("state,a,b", [((2, 2, 2, 3, 0), NORTH, NOOP), ((2, 2, 2, 3, 1), NORTH, NOOP),
                                       ((2, 3, 2, 2, 0), NOOP, NORTH), ((2, 3, 2, 2, 1), NOOP, NORTH)])
def test_collision_through_slip(state, a, b):                            # :175-196
    nxt, _, _ = _one_step_batch(state, a, b, seed=32)
    same_cells = (nxt[:, :4] == np.array(state[:4])).all(1).mean()
    assert np.isclose(same_cells, 0.1, atol=0.02)


def test_no_slip_on_stand():                                             # :198-211
    nxt, _, _ = _one_step_batch((1, 2, 3, 4, 0), NOOP, NOOP, seed=33)
    assert (nxt == np.array((1, 2, 3, 4, 0))).all()


def test_possession_unchanged_without_collision_and_render():            # :61-81
    import gym_soccer_littman94_amd as gsa
    env = gsa.SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.2)
    for p in (0, 1):
        env.reset(); env.state = (1, 1, 3, 3, p)
        env.step({'player_a': EAST, 'player_b': WEST})
        assert env.state[4] == p
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        env.render()
    out = buf.getvalue()
    assert "Player A position" in out and "Player B position" in out and "Ball possession" in out and "Last actions" in out


# ---- test_general.py ---------------------------------------------------------------------------------------
def _check_start_state(env, state):
    row_a, col_a, row_b, col_b, possession = state
    assert col_a == 2 and col_b == env.width - 3
    if len(env.goal_rows) % 2 == 0:
        mid = len(env.goal_rows) // 2
        valid = [env.goal_rows[mid - 1], env.goal_rows[mid]]
        assert row_a in valid and row_b in valid and row_a != row_b
    else:
        middle = env.goal_rows[len(env.goal_rows) // 2]
        assert row_a == middle and row_b == middle
    assert possession in [0, 1]


@pytest.mark.parametrize("width,height", PITCHES)
def test_initial_state_distribution_and_sampling(width, height):         # :12-59, :100-156
    import gym_soccer_littman94_amd as gsa
    env = gsa.SoccerSimultaneousEnv(width=width, height=height)
    assert abs(sum(p for p, _ in env.isd) - 1.0) < 1e-6
    assert all(abs(p - env.isd[0][0]) < 1e-6 for p, _ in env.isd)
    for _, st in env.isd:
        _check_start_state(env, st)
    expected_states = 4 if len(env.goal_rows) % 2 == 0 else 2
    assert len(env.isd) == expected_states
    n_samples, counts = 10000, {}
    for _ in range(n_samples):
        env.reset()
        counts[env.state] = counts.get(env.state, 0) + 1
    assert len(counts) == expected_states
    for st, c in counts.items():
        _check_start_state(env, st)
        assert np.isclose(c, n_samples / expected_states, rtol=0.1)
    observed = np.array(list(counts.values()))
    assert np.std(observed) / np.mean(observed) < 0.05


@pytest.mark.parametrize("width,height", PITCHES)
def test_env_P_structure(width, height):                                 # :61-98 (11x7: the reference's constructor needs 120 s)
    import gym_soccer_littman94_amd as gsa
    env = gsa.SoccerSimultaneousEnv(width=width, height=height)
    P = env.P
    assert isinstance(P, dict) and set(P.keys()) == set(range(len(P)))
    valid_actions = set(P[0].keys())
    for state, actions in P.items():
        assert isinstance(actions, dict) and set(actions.keys()) == valid_actions
        for action, transitions in actions.items():
            assert isinstance(transitions, list)
            for prob, next_state, reward, done in transitions:
                assert 0 <= prob <= 1
                assert isinstance(next_state, int) and 0 <= next_state < len(P)
                assert isinstance(reward, (int, float)) and isinstance(done, bool)


@pytest.mark.parametrize("learner", ["player_a", "player_b", None])
def test_single_and_multi_agent_surface(learner):                        # :159-302
    import gym_soccer_littman94_amd as gsa
    from gym_soccer_littman94_amd import spaces
    n_states, n_actions = 761, 5
    policy = {s: np.random.randint(0, n_actions) for s in range(n_states)}
    kw = {} if learner is None else ({"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy})
    env = gsa.SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.2, **kw)
    assert env.multiagent == (learner is None)
    agents = ["player_a", "player_b"] if learner is None else [learner]
    absent = [a for a in ("player_a", "player_b") if a not in agents]
    assert isinstance(env.observation_space, spaces.Dict) and isinstance(env.action_space, spaces.Dict)
    for a in agents:
        assert env.observation_space[a].n == n_states and env.action_space[a].n == n_actions
    for a in absent:
        assert a not in env.observation_space and a not in env.action_space
    obs, info = env.reset()
    for d in (obs, info):
        assert isinstance(d, dict) and all(a in d for a in agents) and not any(a in d for a in absent)
    assert all(0 <= obs[a] < n_states for a in agents)
    out = env.step({a: np.random.randint(0, n_actions) for a in agents})
    for d in out:
        assert isinstance(d, dict) and all(a in d for a in agents) and not any(a in d for a in absent)
    obs, reward, terminated, truncated, info = out
    assert all(0 <= obs[a] < n_states for a in agents)
    assert all(isinstance(reward[a], float) and isinstance(terminated[a], bool) and isinstance(truncated[a], bool) for a in agents)


@pytest.mark.parametrize("learner,opponent", [("player_a", "stand"), ("player_a", "random"), ("player_b", "stand"), ("player_b", "random")])
def test_value_iteration_best_response_wins(learner, opponent):          # :304-458
    import gym_soccer_littman94_amd as gsa
    from gym_soccer_littman94_amd.planners import value_iteration
    from gym_soccer_littman94_amd.policies import get_random_policy, get_stand_policy
    policy = get_stand_policy(761) if opponent == "stand" else get_random_policy(761, 5, seed=42)
    kw = {"player_b_policy": policy} if learner == "player_a" else {"player_a_policy": policy}
    env = gsa.SoccerSimultaneousEnv(width=5, height=4, slip_prob=0.2, **kw)
    pi, V, Q, cc = value_iteration(env, theta=1e-10, discount_factor=0.99)        # one kernel on the device
    wins, n_episodes = 0, 1000
    for _ in range(n_episodes):
        obs, _ = env.reset()
        done = False
        while not done:
            obs, reward, terminated, truncated, _ = env.step({learner: pi[obs[learner]]})
            done = terminated[learner] or truncated[learner]
            if terminated[learner] and reward[learner] > 0:
                wins += 1
    win_rate = wins / n_episodes
    assert (win_rate == 1.0) if opponent == "stand" else (win_rate > 0.95), win_rate
