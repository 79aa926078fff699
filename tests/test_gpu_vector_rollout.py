"""-m gpu: VectorSoccerEnv.rollout(T, ...) is DEFINED as T successive step() calls stacked over T (the per-step return tuple of
gym_soccer/envs/soccer_simultaneous_env.py:397-408, vectorised): same ticks, same auto-reset convention, same infos at every step.
Twin environments (same seed, same lanes) — one stepped T times, one rolled out — must agree in every value, in both io modes,
with and without slip, in multi-agent and in single-agent (fixed-opponent) mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gym_soccer_littman94_amd import SoccerBatch, VectorSoccerEnv


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def _policy(nS, seed):
    return np.random.default_rng(seed).integers(0, 5, size=nS).astype(np.int8)


def _twins(n, slip, mode, io, **kw):
    pol = {}
    if mode == "learner_a":
        pol = {"player_b_policy": _policy(761, 3)}
    elif mode == "learner_b":
        pol = {"player_a_policy": _policy(761, 4)}
    mk = lambda: VectorSoccerEnv(n, slip_prob=slip, seed=77, io=io, lane_offset=4 * 1000, **pol, **kw)
    return mk(), mk()


@pytest.mark.parametrize("infos", ["last", "all"])
@pytest.mark.parametrize("slip", [0.0, 0.2])
@pytest.mark.parametrize("mode", ["multiagent", "learner_a", "learner_b"])
def test_device_rollout_equals_T_step_calls_on_65536_lanes(slip, mode, infos):
    import torch
    n, T = 65536, 130                      # > 100 steps: every lane truncates at least once, goals auto-reset in between
    e1, e2 = _twins(n, slip, mode, "device")
    e1.reset(); e2.reset()
    ags = e1.return_agent
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    acts = {ag: torch.randint(0, 5, (T, n), dtype=torch.int8, device="cuda", generator=g) for ag in ags}
    # two rollouts back to back (the second starts mid-episode and re-uses the buffers), against 2 T single steps
    for part in range(2):
        O, R, TE, TR, I = e1.rollout(T, acts, infos=infos)
        assert set(O) == set(R) == set(TE) == set(TR) == set(ags)
        every = infos == "all"
        shape = (T, n) if every else (n,)
        for ag in ags:
            assert O[ag].shape == (T, n) and R[ag].dtype == torch.float32 and TE[ag].dtype == torch.bool
            assert I["final_observation"][ag].shape == shape and I[ag]["p"].shape == shape and I["_final_observation"].shape == shape
        for k in range(T):
            o, r, te, tr, i = e2.step({ag: acts[ag][k] for ag in ags})
            for ag in ags:
                assert torch.equal(O[ag][k], o[ag]), (part, k, ag)
                assert torch.equal(R[ag][k], r[ag]), (part, k, ag)
                assert torch.equal(TE[ag][k], te[ag]) and torch.equal(TR[ag][k], tr[ag]), (part, k, ag)
                if every:                       # every step's infos, as after that step()
                    assert torch.equal(I[ag]["p"][k], i[ag]["p"]), (part, k, ag)
                    assert torch.equal(I["final_observation"][ag][k], i["final_observation"][ag]), (part, k, ag)
            if every:
                assert torch.equal(I["_final_observation"][k], i["_final_observation"]), (part, k)
            assert torch.equal(I["reward_int8"][k], e2.reward_int8)
        if every:
            fin_k = I["final_observation"][ags[0]]
            assert bool((fin_k != O[ags[0]]).any()), "no step reported a final observation that differs from the post-reset one"
        else:                                   # the last step's infos, as after the T-th step()
            for ag in ags:
                assert torch.equal(I[ag]["p"], i[ag]["p"]) and torch.equal(I["final_observation"][ag], i["final_observation"][ag])
            assert torch.equal(I["_final_observation"], i["_final_observation"])
        assert bool(TR[ags[0]].any()) and bool(TE[ags[0]].any())
    s1, s2 = e1.get_state(), e2.get_state()
    for key in s1:
        np.testing.assert_array_equal(s1[key], s2[key])
    assert e1.batch.tick == e2.batch.tick == 1 + 2 * T
    np.testing.assert_array_equal(e1.episode_histogram(), e2.episode_histogram())
    e1.close(); e2.close()


@pytest.mark.parametrize("slip", [0.0, 0.2])
@pytest.mark.parametrize("mode", ["multiagent", "learner_b"])
def test_numpy_rollout_equals_T_step_calls(slip, mode):
    n, T = 4096 + 3, 117                   # a ragged tail: the per-lane kernels behind the byte-parallel ones
    e1, e2 = _twins(n, slip, mode, "numpy")
    e1.reset(); e2.reset()
    ags = e1.return_agent
    rng = np.random.default_rng(8)
    acts = {ag: rng.integers(0, 5, size=(T, n)) for ag in ags}
    O, R, TE, TR, I = e1.rollout(T, acts, infos="all")
    for k in range(T):
        o, r, te, tr, i = e2.step({ag: acts[ag][k] for ag in ags})
        for ag in ags:
            np.testing.assert_array_equal(O[ag][k], o[ag]); np.testing.assert_array_equal(R[ag][k], r[ag])
            np.testing.assert_array_equal(TE[ag][k], te[ag]); np.testing.assert_array_equal(TR[ag][k], tr[ag])
            np.testing.assert_array_equal(I[ag]["p"][k], i[ag]["p"]); np.testing.assert_array_equal(I["final_observation"][ag][k], i["final_observation"][ag])
        np.testing.assert_array_equal(I["_final_observation"][k], i["_final_observation"])
    for ag in ags:
        assert R[ag].dtype == np.float32 and TE[ag].dtype == np.bool_
        np.testing.assert_array_equal(I[ag]["p"][-1], i[ag]["p"])
        np.testing.assert_array_equal(I["final_observation"][ag][-1], i["final_observation"][ag])
    np.testing.assert_array_equal(I["_final_observation"][-1], i["_final_observation"])
    assert I["final_observation"][ags[0]].shape == (T, n) and I["_final_observation"].shape == (T, n)
    s1, s2 = e1.get_state(), e2.get_state()
    for key in s1:
        np.testing.assert_array_equal(s1[key], s2[key])
    # the default, infos="last" (T - 1 fused steps + one full step), on a third twin: the same trajectories, the T-th step's infos
    e3, e4 = _twins(n, slip, mode, "numpy")
    e3.reset(); e4.close()
    O3, R3, TE3, TR3, I3 = e3.rollout(T, acts)
    for ag in ags:
        np.testing.assert_array_equal(O3[ag], O[ag]); np.testing.assert_array_equal(R3[ag], R[ag])
        np.testing.assert_array_equal(TE3[ag], TE[ag]); np.testing.assert_array_equal(TR3[ag], TR[ag])
        np.testing.assert_array_equal(I3[ag]["p"], I[ag]["p"][-1])
        np.testing.assert_array_equal(I3["final_observation"][ag], I["final_observation"][ag][-1])
    np.testing.assert_array_equal(I3["_final_observation"], I["_final_observation"][-1])
    np.testing.assert_array_equal(I3["reward_int8"], I["reward_int8"])
    e1.close(); e2.close(); e3.close()


def test_one_step_rollout_and_argument_checks():
    e1, e2 = _twins(1024, 0.2, "multiagent", "numpy")
    with pytest.raises(AssertionError, match="reset the environment"):
        e1.rollout(3, {"player_a": np.zeros((3, 1024), int), "player_b": np.zeros((3, 1024), int)})
    e1.reset(); e2.reset()
    a = np.random.default_rng(1).integers(0, 5, size=(1, 2, 1024))
    O, R, TE, TR, I = e1.rollout(1, {"player_a": a[:, 0], "player_b": a[:, 1]})          # infos="last": the default
    o, r, te, tr, i = e2.step({"player_a": a[0, 0], "player_b": a[0, 1]})
    np.testing.assert_array_equal(O["player_a"][0], o["player_a"]); np.testing.assert_array_equal(R["player_b"][0], r["player_b"])
    np.testing.assert_array_equal(I["player_a"]["p"], i["player_a"]["p"])
    np.testing.assert_array_equal(I["final_observation"]["player_b"], i["final_observation"]["player_b"])
    np.testing.assert_array_equal(I["_final_observation"], i["_final_observation"])
    with pytest.raises(AssertionError, match="'last', 'all' or 'none'"):
        e1.rollout(2, {"player_a": np.zeros((2, 1024), int), "player_b": np.zeros((2, 1024), int)}, infos="some")
    with pytest.raises(AssertionError, match="0..4"):
        e1.rollout(2, {"player_a": np.full((2, 1024), 5), "player_b": np.zeros((2, 1024), int)})
    with pytest.raises(AssertionError, match="both 'player_a' and 'player_b'|length 2"):
        e1.rollout(2, {"player_a": np.zeros((2, 1024), int)})
    with pytest.raises(AssertionError, match=r"\[n_steps, num_envs\]"):
        e1.rollout(2, {"player_a": np.zeros((3, 1024), int), "player_b": np.zeros((3, 1024), int)})
    e1.close(); e2.close()


@pytest.mark.parametrize("mixed", [False, True])
def test_sampled_rollout_is_batched_rollout_with_in_kernel_actions(mixed):
    """sample_actions=True: the env's rollout is batched_rollout's in-kernel sampling (uniform, or from [nS, 5] mixed policies —
    BASELINE config 5), i.e. what a twin SoccerBatch with the same seed and lanes produces."""
    import torch
    n, T = 8192, 100
    env = VectorSoccerEnv(n, slip_prob=0.2, seed=11, io="device")
    b = SoccerBatch(n, 5, 4, 0.2, seed=11, autoreset=True)
    env.reset(); b.reset()
    mp, kw = None, {}
    if mixed:
        rng = np.random.default_rng(94)
        mp = {"player_a": rng.dirichlet(np.ones(5) * 0.7, size=env.nS), "player_b": rng.dirichlet(np.ones(5) * 0.7, size=env.nS)}
        kw = {"mix_a": b.alloc((env.nS, 4), np.uint16).upload(SoccerBatch.mixed_policy_thresholds(mp["player_a"])),
              "mix_b": b.alloc((env.nS, 4), np.uint16).upload(SoccerBatch.mixed_policy_thresholds(mp["player_b"]))}
    O, R, TE, TR, I = env.rollout(T, sample_actions=True, mixed_policies=mp, infos="all")
    obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8); term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
    b.rollout(T, sample_actions=True, obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n, **kw)
    np.testing.assert_array_equal(_np(O["player_a"]).view(np.uint16), obs.download())
    np.testing.assert_array_equal(_np(I["reward_int8"]), rew.download())
    np.testing.assert_array_equal(_np(R["player_b"]), -rew.download().astype(np.float32))
    np.testing.assert_array_equal(_np(TE["player_a"]), term.download().view(np.bool_))
    np.testing.assert_array_equal(_np(TR["player_b"]), trunc.download().view(np.bool_))
    np.testing.assert_array_equal(_np(I["_final_observation"]), (term.download() | trunc.download()).view(np.bool_))
    # the sampled rollout reports every step's final observation too: where nothing ended it is the observation itself
    fin = _np(I["final_observation"]["player_a"]).view(np.uint16); same = ~(term.download() | trunc.download()).view(np.bool_)
    np.testing.assert_array_equal(fin[same], obs.download()[same])
    assert (fin[~same] != obs.download()[~same]).any()
    np.testing.assert_array_equal(env.episode_histogram(), b.stats()[0])
    assert int(_np(TE["player_a"]).sum()) > 0
    env.close(); b.close()


def test_lean_env_rollout_writes_no_info_trajectories():
    import torch
    n, T = 4096, 20
    e1 = VectorSoccerEnv(n, slip_prob=0.2, seed=3, io="device", info=False)
    e2 = VectorSoccerEnv(n, slip_prob=0.2, seed=3, io="device", info=False)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = {ag: torch.randint(0, 5, (T, n), dtype=torch.int8, device="cuda", generator=g) for ag in ("player_a", "player_b")}
    O, R, TE, TR, I = e1.rollout(T, acts, infos="all")                  # an info=False env has none to give: "none" whatever is asked
    assert "final_observation" not in I and "player_a" not in I and I["_final_observation"].shape == (T, n)
    for k in range(T):
        o, r, te, tr, i = e2.step({ag: acts[ag][k] for ag in acts})
        assert torch.equal(O["player_a"][k], o["player_a"]) and torch.equal(R["player_b"][k], r["player_b"])
        assert torch.equal(I["_final_observation"][k], i["_final_observation"])
    e1.close(); e2.close()
