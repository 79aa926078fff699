"""CPU tests of the host-side logic that needs no device: spaces stand-ins, registration table,
mixed-policy threshold conversion, the C oracle's action samplers, and the closed-form state numbering
against the reference's tables."""
import glob
import os

import numpy as np
import pytest

import gym_soccer_littman94_amd as gsa
from gym_soccer_littman94_amd import spaces
from gym_soccer_littman94_amd.core import SoccerBatch
from gym_soccer_littman94_amd.registration import ENV_SPECS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_spaces_behave_like_gym_spaces():
    d = spaces.Dict({'player_a': spaces.Discrete(761), 'player_b': spaces.Discrete(5)})
    assert d['player_a'].n == 761 and 'player_b' in d and len(d) == 2
    x = d.sample()
    assert set(x) == {'player_a', 'player_b'} and 0 <= x['player_a'] < 761 and d.contains(x)
    assert not d.contains({'player_a': 761, 'player_b': 0}) and not d.contains({'player_a': 0})
    m = spaces.MultiDiscrete(np.full(16, 5))
    s = m.sample()
    assert s.shape == (16,) and m.contains(s) and not m.contains(s + 5)


def test_registration_table_matches_the_reference_stub():
    # gym_soccer/__init__.py:5-12 (commented out there): id, entry point class, kwargs
    spec = ENV_SPECS["SoccerSimultaneous-v0"]
    assert spec["entry_point"].endswith(":SoccerSimultaneousEnv")
    assert spec["kwargs"] == {"width": 5, "height": 4, "slip_prob": 0.2, "player_a_policy": None, "player_b_policy": None}
    assert ENV_SPECS["SoccerLittman94-v0"]["kwargs"]["slip_prob"] == 0.0
    with pytest.raises(KeyError):
        gsa.make("NoSuchEnv-v0")
    assert isinstance(gsa.register_all(), list)


def test_mixed_policy_thresholds_are_exact_and_monotone():
    rng = np.random.default_rng(0)
    p = rng.dirichlet(np.ones(5), size=761)
    t = SoccerBatch.mixed_policy_thresholds(p)
    assert t.dtype == np.uint16 and t.shape == (761, 4) and (np.diff(t.astype(int), axis=1) >= 0).all()
    # a deterministic policy maps to thresholds that select exactly that action for every 15-bit draw
    det = np.eye(5)[rng.integers(0, 5, size=761)]
    td = SoccerBatch.mixed_policy_thresholds(det).astype(np.int64)
    for draw in (0, 1, 16384, 32767):
        np.testing.assert_array_equal((draw >= td).sum(1), det.argmax(1))
    with pytest.raises(AssertionError):
        SoccerBatch.mixed_policy_thresholds(np.ones((761, 5)))
    # empirical frequencies of the sampler follow the table
    probs = np.array([[0.1, 0.2, 0.3, 0.15, 0.25]])
    th = SoccerBatch.mixed_policy_thresholds(probs).astype(np.int64)[0]
    draws = np.arange(32768)
    freq = np.bincount((draws[:, None] >= th[None, :]).sum(1), minlength=5) / 32768
    np.testing.assert_allclose(freq, probs[0], atol=4e-5)


def test_oracle_action_samplers_are_uniform_and_lane_stable():
    from oracle.oracle import Oracle
    o = Oracle(5, 4, 0.0, n=200000, seed=3, lane_offset=8)
    a, b = o.sample_actions(tick=5)
    for x in (a, b):
        f = np.bincount(x, minlength=5) / x.size
        assert x.min() >= 0 and x.max() <= 4 and np.abs(f - 0.2).max() < 0.01
    # lane g's draw depends on the global lane id only, not on where the shard starts
    o2 = Oracle(5, 4, 0.0, n=1000, seed=3, lane_offset=8 + 777)
    a2, b2 = o2.sample_actions(tick=5)
    np.testing.assert_array_equal(a2, a[777:1777]); np.testing.assert_array_equal(b2, b[777:1777])
    a3, _ = o.sample_actions(tick=6)
    assert (a3 != a).mean() > 0.7


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "table_*_s0.npz"))))
def test_closed_form_state_numbering_matches_reference_tables(path):
    """obs = 1 + 2*(iA*(NI-1) + iB - (iB > iA)) + p over interior-cell indices (csrc/soccer_rules.hpp
    self-checks its table against the same formula at create time)."""
    g = np.load(path)
    W = int(g["width"]) + 2; H = int(g["height"]); NI = H * (W - 2)
    lut = g["lut"]; kind = g["kind"]
    assert int(g["nS"]) == NI * (NI - 1) * 2 + 1
    idx = np.arange(lut.size)
    p = idx & 1; r = idx >> 1
    yb = r % W; r //= W; xb = r % H; r //= H; ya = r % W; xa = r // W
    live = kind == 1
    ia = xa * (W - 2) + ya - 1; ib = xb * (W - 2) + yb - 1
    want = 1 + 2 * (ia * (NI - 1) + ib - (ib > ia)) + p
    np.testing.assert_array_equal(lut[live], want[live])
    interior = (ya > 0) & (ya < W - 1) & (yb > 0) & (yb < W - 1) & ~((xa == xb) & (ya == yb))
    np.testing.assert_array_equal(live, interior)
    assert (lut[kind == 2] == 0).all() and (lut[kind == 0] == 0xFFFF).all()


def test_threshold_draw_fixture_is_what_it_claims():
    """tests/golden/threshold_draws.json (tools/find_threshold_draws.py): every recorded draw is the spec's Philox
    word of that (lane, tick) and sits on or next to an integer slip threshold c = ceil(running sum * 2^30 - 1/2)
    (include/soccer_hip.h: u = (m + 1/2) * 2^-30, so "sum <= u" is "m >= c")."""
    import json
    import os
    import numpy as np
    from oracle.oracle import philox4x32_10
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "threshold_draws.json")))
    seed = d["seed"]
    assert len(d["hits"]) >= 50 and {h["slip"] for h in d["hits"]} >= {0.1, 0.2, 0.3, 0.5}
    # draws exactly ON slip 0.1's mathematically dyadic threshold 27/32 (0.81 + 3 * 0.01125): where round 2 needed a float64 walk
    assert sum(1 for h in d["hits"] if h["slip"] == 0.1 and h["m"] == 27 * 2 ** 25) >= 3
    cls = [0, 1, 1, 2, 2, 3, 3, 3, 3]
    for h in d["hits"]:
        q = h["lane"] >> 2
        w = philox4x32_10([q & 0xffffffff, q >> 32, h["tick"], 0], [seed & 0xffffffff, seed >> 32])
        assert int(w[h["lane"] & 3]) >> 2 == h["m"]
        s = np.float64(h["slip"]); om = np.float64(1) - s
        wt = [om * om, (om * s) * 0.5, (s * om) * 0.5, (s * s) * 0.25]
        acc, th = np.float64(0), {}
        for c in range(9):
            S = acc; acc = acc + wt[cls[c]]
            th[("end", c)] = acc; th[("two", c)] = S + wt[cls[c]] * 0.5
            t = S + wt[cls[c]] * 0.25; th[("four1", c)] = t
            t = t + wt[cls[c]] * 0.25; th[("four2", c)] = t
            t = t + wt[cls[c]] * 0.25; th[("four3", c)] = t
        for name, c in h["thresholds"]:
            cb = int(np.ceil(float(th[(name, c)]) * 2.0 ** 30 - 0.5))
            assert h["m"] in (cb - 1, cb, cb + 1)


def test_install_as_gym_soccer_aliases_the_reference_import_names():
    import sys
    import gym_soccer_littman94_amd as gsa
    assert "gym_soccer" not in sys.modules
    gsa.install_as_gym_soccer()
    try:
        from gym_soccer.envs import SoccerSimultaneousEnv
        from gym_soccer.envs.soccer_simultaneous_env import SoccerSimultaneousEnv as E2
        from gym_soccer.utils.planners import modified_policy_iteration, policy_iteration, value_iteration
        from gym_soccer.utils.policies import get_random_policy, get_stand_policy
        assert SoccerSimultaneousEnv is gsa.SoccerSimultaneousEnv is E2
        assert value_iteration is gsa.planners.value_iteration and policy_iteration and modified_policy_iteration
        assert get_stand_policy(3) == {0: 0, 1: 0, 2: 0} and len(get_random_policy(5, 5, 1)) == 5
        gsa.install_as_gym_soccer()                  # idempotent over its own alias
    finally:
        for k in [k for k in sys.modules if k == "gym_soccer" or k.startswith("gym_soccer.")]:
            del sys.modules[k]


def test_lazy_result_dicts_of_the_device_vector_env():
    """VectorSoccerEnv(io="device") hands out dicts whose expensive values (float32 rewards, info["p"],
    "_final_observation") are computed on first access and forgotten at the next step — plain-dict behaviour otherwise."""
    from gym_soccer_littman94_amd.envs.vector_env import _Lazy, _LazyInfo
    calls = []
    src = {"v": 1}
    d = _Lazy({"player_a": lambda: calls.append("a") or src["v"], "player_b": lambda: calls.append("b") or -src["v"]})
    assert isinstance(d, dict) and "player_a" in d and "nobody" not in d and len(d) == 2 and list(d) == ["player_a", "player_b"]
    assert calls == []
    assert d["player_a"] == 1 and d["player_a"] == 1 and calls == ["a"]              # computed once, then cached
    assert d.get("player_b") == -1 and d.get("nobody", 7) == 7
    assert dict(d.items()) == {"player_a": 1, "player_b": -1} and sorted(d.values()) == [-1, 1]
    src["v"] = 5; d.invalidate()
    assert d["player_a"] == 5 and calls == ["a", "b", "a"]
    with pytest.raises(KeyError):
        d["nobody"]
    info = _LazyInfo(lambda: "p-array")
    top = _Lazy({"_final_observation": lambda: "mask"}, {"player_a": info, "final_observation": {"player_a": "fin"}})
    assert set(top.keys()) == {"player_a", "final_observation", "_final_observation"}
    assert top["player_a"]["p"] == "p-array" and top["final_observation"]["player_a"] == "fin" and top["_final_observation"] == "mask"
    top.invalidate()
    assert top["player_a"] is info and "_final_observation" in top


def test_numpy_philox_known_answers_and_draw_convention():
    """tests/philox_np.py (used by tests/test_gpu_fixture_pin.py instead of the oracle): Random123's kat_vectors for
    philox4x32 with 10 rounds, and the ABI-3 bits -> uniform convention against its definition in include/soccer_hip.h."""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from philox_np import lane_words, philox4x32_10, step_draws
    kat = [(([0, 0, 0, 0], [0, 0]), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           (([0xffffffff] * 4, [0xffffffff] * 2), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           (([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for (ctr, key), want in kat:
        got = philox4x32_10([ctr[0]], [ctr[1]], [ctr[2]], [ctr[3]], key[0], key[1])[:, 0]
        assert [int(x) for x in got] == want
    seed, lanes = 0x1234567890abcdef, np.array([0, 1, 2, 3, 4, 7, (1 << 34) + 5], np.uint64)
    for tick in (0, 1, 6, 7, 8, 1234567, (1 << 33) + 3):
        # slip > 0: the tick's own block, 30 + 2 bits of the lane's word
        w = lane_words(seed, lanes, tick)
        for j, g in enumerate(lanes.tolist()):
            q = g >> 2
            blk = philox4x32_10([q & 0xffffffff], [q >> 32], [tick & 0xffffffff], [tick >> 32], seed & 0xffffffff, seed >> 32)[:, 0]
            assert int(w[j]) == int(blk[g & 3])
        u, ur = step_draws(seed, lanes, tick, 0.2)
        np.testing.assert_array_equal(u, ((w >> 2).astype(np.float64) + 0.5) / 2.0 ** 30)
        np.testing.assert_array_equal(ur, ((w & 3).astype(np.float64) + 0.5) / 4.0)
        assert ((u > 0) & (u < 1)).all()
        # slip == 0: eight ticks share the block of tick >> 3; nibble (tick & 7) ^ 1
        w8 = lane_words(seed, lanes, tick >> 3)
        u0, ur0 = step_draws(seed, lanes, tick, 0.0)
        nib = (w8.astype(np.int64) >> (4 * ((tick & 7) ^ 1))) & 15
        np.testing.assert_array_equal(u0, ((nib >> 2) + 0.5) / 4.0)
        np.testing.assert_array_equal(ur0, ((nib & 3) + 0.5) / 4.0)
