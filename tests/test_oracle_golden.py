"""Pins the CPU oracle (oracle/soccer_oracle.c) against fixtures dumped from the real reference
(tests/golden/make_golden.py): complete transition tables row-for-row with list order and exact
float64 probabilities, injected-uniform replays through the reference's step()/reset(), and
MT19937-driven trajectories."""
import glob
import os
import sys

import numpy as np
import pytest

from oracle.oracle import Oracle, philox4x32_10

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
from make_golden import table_digest          # the digest's definition lives next to the script that dumped it (imports nothing of the reference)
TABLES = sorted(glob.glob(os.path.join(GOLDEN, "table_*.npz")))
REPLAYS = sorted(glob.glob(os.path.join(GOLDEN, "replay_*.npz")))
RESETS = sorted(glob.glob(os.path.join(GOLDEN, "reset_*.npz")))
TRAJS = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
DIGESTS = sorted(glob.glob(os.path.join(GOLDEN, "digest_*.npz")))


def _ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


def test_fixtures_present():
    assert len(TABLES) >= 8 and len(REPLAYS) >= 8 and len(RESETS) >= 4 and len(TRAJS) >= 4
    # round 4: BASELINE config 1 at its stated length, and the reference's largest parametrisation with slip
    names = {os.path.basename(p) for p in TRAJS + DIGESTS + REPLAYS + RESETS}
    assert {"traj_5x4_s0_seed0_10k.npz", "traj_5x4_s0p2_seed0_10k.npz", "digest_11x7_s0p2.npz", "replay_11x7_s0p2.npz",
            "reset_11x7_s0p2.npz", "digest_5x4_s0p2.npz"} <= names


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert [hex(x) for x in philox4x32_10([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                          [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


@pytest.mark.parametrize("path", TABLES + DIGESTS, ids=_ids(TABLES + DIGESTS))
def test_state_classification_and_isd(path):
    g = np.load(path)
    o = Oracle(int(g["width"]), int(g["height"]), float(g["slip"]))
    lut, kind, gv, isd, isdp = o.tables()
    assert o.nS == int(g["nS"])
    np.testing.assert_array_equal(lut, g["lut"])
    np.testing.assert_array_equal(kind, g["kind"])
    np.testing.assert_array_equal(gv, g["goal_value"])
    np.testing.assert_array_equal(isd, g["isd_states"])
    np.testing.assert_array_equal(isdp, g["isd_probs"])


@pytest.mark.parametrize("path", TABLES, ids=_ids(TABLES))
def test_transition_table_row_for_row(path):
    g = np.load(path)
    o = Oracle(int(g["width"]), int(g["height"]), float(g["slip"]))
    rows, prob = g["rows"], g["prob"]
    # rows: xa,ya,xb,yb,p, aa,ab, k, nxa,nya,nxb,nyb,np, reward, done — grouped by (state, ja), k ascending
    starts = np.flatnonzero(rows[:, 7] == 0)
    ends = np.append(starts[1:], len(rows))
    n_keys = 0
    for s, e in zip(starts, ends):
        st, aa, ab = rows[s, :5], int(rows[s, 5]), int(rows[s, 6])
        p, ns, r, d = o.transitions(st, aa, ab)
        assert len(p) == e - s, (st, aa, ab)
        assert np.array_equal(p, prob[s:e]), (st, aa, ab, p, prob[s:e])      # bit-exact float64
        assert np.array_equal(ns, rows[s:e, 8:13]), (st, aa, ab)
        assert np.array_equal(r, rows[s:e, 13]) and np.array_equal(d, rows[s:e, 14]), (st, aa, ab)
        n_keys += 1
    # every non-unreachable tuple has all 25 joint actions
    assert n_keys == 25 * int(np.count_nonzero(g["kind"]))
    # and the oracle has no key the reference lacks
    W = int(g["width"]) + 2; H = int(g["height"])
    unreachable = np.flatnonzero(g["kind"] == 0)[:50]
    for f in unreachable:
        p_ = f & 1; f //= 2; yb = f % W; f //= W; xb = f % H; f //= H; ya = f % W; xa = f // W
        with pytest.raises(KeyError):
            o.transitions([xa, ya, xb, yb, p_], 0, 0)


def test_digest_definition_on_a_table_that_is_also_stored_in_full():
    """digest_5x4_s0p2.npz was written by the same reference run that table_5x4_s0p2.npz restates row by row: the digest of the
    stored rows must be the stored digest (so a digest match on 11x7 means what a row-for-row match means here)."""
    g = np.load(os.path.join(GOLDEN, "table_5x4_s0p2.npz")); d = np.load(os.path.join(GOLDEN, "digest_5x4_s0p2.npz"))
    sha, per = table_digest(g["rows"], g["prob"])
    assert sha == d["sha256"].item().decode() and len(g["rows"]) == int(d["n_rows"])
    np.testing.assert_array_equal(per, d["tuple_digest"])


@pytest.mark.parametrize("path", DIGESTS, ids=_ids(DIGESTS))
def test_transition_table_digest(path):
    """Tables too large to commit (11x7 with slip: 3.4 M rows): the oracle's complete table, dumped in the reference's own
    iteration order, must hash to what the reference's table hashed to (gym_soccer/envs/soccer_simultaneous_env.py:167-293,
    sizes of gym_soccer/tests/test_general.py:5-11).  A mismatch is localised by the per-tuple digests."""
    d = np.load(path)
    o = Oracle(int(d["width"]), int(d["height"]), float(d["slip"]))
    rows, prob = o.dump_table()
    assert len(rows) == int(d["n_rows"])
    lens = np.diff(np.append(np.flatnonzero(rows[:, 7] == 0), len(rows)))
    np.testing.assert_array_equal(np.bincount(lens, minlength=37), d["list_length_hist"])
    sha, per = table_digest(rows, prob)
    bad = np.flatnonzero(per != d["tuple_digest"])
    assert bad.size == 0, "first differing state tuple (in order of appearance): %d" % bad[0]
    assert sha == d["sha256"].item().decode()


@pytest.mark.parametrize("path", REPLAYS, ids=_ids(REPLAYS))
def test_replay_through_reference_step(path):
    g = np.load(path)
    n = len(g["u"])
    o = Oracle(int(g["width"]), int(g["height"]), float(g["slip"]), n=n)
    st = g["state"]
    o.set_state(st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], t=g["t"], needs_reset=np.zeros(n, np.uint8))
    out = o.step(g["action"][:, 0], g["action"][:, 1], u_step=g["u"])
    ns = g["next_state"]
    np.testing.assert_array_equal(o.row_a, ns[:, 0]); np.testing.assert_array_equal(o.col_a, ns[:, 1])
    np.testing.assert_array_equal(o.row_b, ns[:, 2]); np.testing.assert_array_equal(o.col_b, ns[:, 3])
    np.testing.assert_array_equal(o.poss & 1, ns[:, 4])
    np.testing.assert_array_equal(out["obs"], g["obs"])
    np.testing.assert_array_equal(out["reward"].astype(np.float64), g["reward_a"])
    np.testing.assert_array_equal(-out["reward"].astype(np.float64), g["reward_b"])
    np.testing.assert_array_equal(out["terminated"], g["terminated"])
    np.testing.assert_array_equal(out["truncated"], g["truncated"])
    np.testing.assert_array_equal((o.poss >> 1) & 1, g["needs_reset"])
    np.testing.assert_array_equal(np.round(out["prob"], 2), g["p"])
    assert o.misuse == 0


@pytest.mark.parametrize("path", RESETS, ids=_ids(RESETS))
def test_reset_vectors(path):
    g = np.load(path)
    n = len(g["u"])
    o = Oracle(int(g["width"]), int(g["height"]), float(g["slip"]) if "slip" in g.files else 0.0, n=n)
    obs = o.reset(u_reset=g["u"])
    st = g["state"]
    np.testing.assert_array_equal(obs, g["obs"])
    np.testing.assert_array_equal(o.row_a, st[:, 0]); np.testing.assert_array_equal(o.col_a, st[:, 1])
    np.testing.assert_array_equal(o.row_b, st[:, 2]); np.testing.assert_array_equal(o.col_b, st[:, 3])
    np.testing.assert_array_equal(o.poss, st[:, 4])          # needs_reset bit cleared
    assert not o.t.any()


@pytest.mark.parametrize("path", TRAJS, ids=_ids(TRAJS))
def test_mt19937_trajectories_with_recorded_uniforms(path):
    g = np.load(path)
    o = Oracle(5, 4, float(g["slip"]), n=1)
    assert o.reset(u_reset=[g["u_first_reset"]])[0] == g["first_obs"]
    for k in range(len(g["obs"])):
        if g["reset_before"][k]:
            assert (o.poss[0] >> 1) & 1
            assert o.reset(u_reset=[g["u_reset"][k]])[0] == g["reset_obs"][k]
        out = o.step([g["actions"][k, 0]], [g["actions"][k, 1]], u_step=[g["u_step"][k]])
        assert out["obs"][0] == g["obs"][k], k
        assert out["reward"][0] == g["reward_a"][k] and -float(out["reward"][0]) == g["reward_b"][k]
        assert out["terminated"][0] == g["terminated"][k] and out["truncated"][0] == g["truncated"][k]
        assert np.round(out["prob"][0], 2) == g["p"][k]
        assert tuple(g["state"][k]) == (o.row_a[0], o.col_a[0], o.row_b[0], o.col_b[0], o.poss[0] & 1)
    assert o.misuse == 0
