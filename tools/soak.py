#!/usr/bin/env python3
"""Randomised soak of the HIP path against the CPU oracle (run on the GPU box): random pitch, slip, seed, lane count,
lane offset, auto-reset, single steps (lean / full; Philox or caller-supplied uniforms) and fused rollouts (streams / sampled / mixed policies / single-agent),
every lane of every step compared (incl. the ABI-2 gym outputs, aligned and not).  Usage: tools/soak.py [seconds [seed]]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from gym_soccer_littman94_amd import SoccerBatch
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
soak_seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())       # printed, so a failing run can be repeated exactly
print("soak seed %d" % soak_seed, flush=True)
rng = np.random.default_rng(soak_seed)
pitches = [(5, 4), (6, 4), (7, 5), (9, 6), (11, 7), (5, 5), (8, 4), (6, 6), (13, 9)]
slips = [0.0, 0.0, 0.2, 0.1, 0.3, 0.5, 0.9, 1.0, 0.05, 1 / 3, 0.25]
t0 = time.time(); rounds = 0; lanes_steps = 0; last_note = t0
while time.time() - t0 < budget:
    w, h = pitches[rng.integers(len(pitches))]; slip = slips[rng.integers(len(slips))]
    n = int(rng.choice([4, 64, 1000, 4096, 4099, 8192 + 4 * int(rng.integers(0, 50))]))
    off = int(rng.integers(0, 1 << 40)) & ~3 if rng.random() < 0.8 else int(rng.integers(0, 1 << 20))
    seed = int(rng.integers(0, 1 << 62)); autoreset = bool(rng.random() < 0.7)
    ms = int(rng.choice([100, 100, 17, 127, 200]))
    b = SoccerBatch(n, w, h, slip, seed=seed, autoreset=autoreset, lane_offset=off, max_steps=ms, step_stats=bool(rng.random() < 0.5),
                    stream_actions=bool(rng.random() < 0.3))           # (both arms of the step kernel's action loads)
    o = Oracle(w, h, slip, n=n, seed=seed, autoreset=autoreset, lane_offset=off, max_steps=ms)
    tag = "pitch %dx%d slip %g n %d off %d seed %d autoreset %s max_steps %d" % (w, h, slip, n, off, seed, autoreset, ms)
    try:
        b.reset(); cur = o.reset()
        full = rng.random() < 0.5
        aa = b.alloc(n, np.int8); ab = b.alloc(n, np.int8)
        obs = b.alloc(n, np.uint16); rew = b.alloc(n, np.int8); te = b.alloc(n, np.uint8); tr = b.alloc(n, np.uint8)
        code = b.alloc(n, np.uint8) if full else None; fin = b.alloc(n, np.uint16) if full else None
        # the gym outputs (ABI 2), each asked for independently and sometimes only 4-byte aligned (byte-I/O fallback)
        gym = full and rng.random() < 0.6
        sh = int(rng.choice([0, 0, 0, 1, 3])) if gym else 0
        rfa = b.alloc(n + 4, np.float32).fill(7) if gym and rng.random() < 0.8 else None
        rfb = b.alloc(n + 4, np.float32).fill(7) if gym and rng.random() < 0.8 else None
        dn = b.alloc(n + 4, np.uint8).fill(7) if gym and rng.random() < 0.8 else None
        # caller-supplied uniforms (the reference-RNG replay path): either stream, sometimes only 8-byte aligned (per-lane kernel),
        # with the values the first-exceeds rule is sensitive to mixed in (dyadic thresholds, 0, 1 - ulp, out of range, NaN)
        expl = rng.random() < 0.3
        ush = int(rng.choice([0, 0, 1])) if expl else 0
        du = b.alloc(n + 2, np.float64) if expl and rng.random() < 0.8 else None
        dr = b.alloc(n + 2, np.float64) if expl and (du is None or rng.random() < 0.7) else None
        edge = np.array([0.0, 0.25, 0.5, 0.75, 0.2499999999999999, 0.4999999999999999, 1.0 - 2.0 ** -53, 2.0 ** -53, 1.0, -0.5, np.nan, 7.0])
        # single-agent single steps (the fixed side acts on the current observation, :187-188), with and without the caller's uniforms
        early_pol = rng.integers(0, 5, size=o.nS).astype(np.int8) if rng.random() < 0.2 else None
        if early_pol is not None:
            b.set_policy("player_b", early_pol)
        for k in range(int(rng.integers(5, 60))):
            a = rng.integers(0, 5, size=(2, n), dtype=np.int8)
            if early_pol is not None:
                a[1] = early_pol[cur]
            aa.upload(a[0]); ab.upload(a[1])
            us = ur = None
            if du is not None:
                us = np.where(rng.random(n) < 0.2, edge[rng.integers(0, len(edge), n)], rng.random(n))
                du.upload(np.concatenate([np.zeros(ush), us, np.zeros(2 - ush)]))
            if dr is not None:
                ur = np.where(rng.random(n) < 0.2, edge[rng.integers(0, len(edge), n)], rng.random(n))
                dr.upload(np.concatenate([np.zeros(ush), ur, np.zeros(2 - ush)]))
            b.step(aa, None if early_pol is not None else ab, obs=obs, reward=rew, terminated=te, truncated=tr, prob_code=code, final_obs=fin,
                   reward_a_f32=None if rfa is None else rfa.ptr + 4 * sh, reward_b_f32=None if rfb is None else rfb.ptr + 4 * sh,
                   finished=None if dn is None else dn.ptr + sh,
                   u_step=None if du is None else du.ptr + 8 * ush, u_reset=None if dr is None else dr.ptr + 8 * ush)
            c = o.step(a[0], a[1], u_step=us, u_reset=ur)
            assert np.array_equal(obs.download(), c["obs"]) and np.array_equal(rew.download(), c["reward"]), "step " + tag
            assert np.array_equal(te.download(), c["terminated"]) and np.array_equal(tr.download(), c["truncated"]), "step flags " + tag
            if full:
                assert np.array_equal(code.download(), c["prob_code"]) and np.array_equal(fin.download(), c["final_obs"]), "step full " + tag
            r32 = c["reward"].astype(np.float32)
            if rfa is not None:
                assert np.array_equal(rfa.download()[sh:sh + n].view(np.uint32), r32.view(np.uint32)), "reward_a_f32 " + tag
            if rfb is not None:
                assert np.array_equal(rfb.download()[sh:sh + n].view(np.uint32), (np.float32(0) - r32).view(np.uint32)), "reward_b_f32 " + tag
            if dn is not None:
                assert np.array_equal(dn.download()[sh:sh + n], c["terminated"] | c["truncated"]), "finished " + tag
            cur = c["obs"]; lanes_steps += n
        if early_pol is not None:
            b.set_policy("player_b", None)
        for arr, fill in ((rfa, 0x07070707), (rfb, 0x07070707), (dn, 7)):       # nothing written outside [sh, sh + n)
            if arr is not None:
                d = arr.download(); d = d.view(np.uint32) if d.dtype == np.float32 else d
                assert (d[:sh] == fill).all() and (d[sh + n:] == fill).all(), "gym output overrun " + tag
        if rng.random() < 0.3:      # masked reset in between
            m = (rng.random(n) < 0.3).astype(np.uint8)
            md = b.alloc(n, np.uint8).upload(m)
            b.reset(mask=md, obs=obs); cur = o.reset(mask=m)
            assert np.array_equal(obs.download(), cur), "masked reset " + tag
        T = int(rng.integers(1, 50)); mode = rng.integers(0, 4)
        O = b.alloc((T, n), np.uint16); R = b.alloc((T, n), np.int8); TE = b.alloc((T, n), np.uint8); TR = b.alloc((T, n), np.uint8)
        kw = dict(obs=O, reward=R, terminated=TE, truncated=TR, out_stride=n)
        # batched_rollout_ex: the per-step final observation / probability code, each asked for independently
        FO = b.alloc((T, n), np.uint16) if rng.random() < 0.5 else None
        CD = b.alloc((T, n), np.uint8) if rng.random() < 0.5 else None
        kw.update(final_obs=FO, prob_code=CD)
        acts = rng.integers(0, 5, size=(T, 2, n), dtype=np.int8)
        mix = None; pol = None
        if mode == 0:
            A = b.alloc((T, n), np.int8).upload(acts[:, 0]); B = b.alloc((T, n), np.int8).upload(acts[:, 1])
            b.rollout(T, A, B, act_stride=n, **kw)
        elif mode == 1:
            b.rollout(T, sample_actions=True, **kw)
        elif mode == 2:
            mix = [SoccerBatch.mixed_policy_thresholds(rng.dirichlet(np.ones(5) * 0.5, size=o.nS)) for _ in range(2)]
            da = b.alloc(mix[0].shape, np.uint16).upload(mix[0]); db = b.alloc(mix[1].shape, np.uint16).upload(mix[1])
            b.rollout(T, sample_actions=True, mix_a=da, mix_b=db, **kw)
        else:
            pol = rng.integers(0, 5, size=o.nS).astype(np.int8)
            b.set_policy("player_b", pol)
            A = b.alloc((T, n), np.int8).upload(acts[:, 0])
            b.rollout(T, A, None, act_stride=n, **kw)
        Oh, Rh, TEh, TRh = O.download(), R.download(), TE.download(), TR.download()
        FOh = None if FO is None else FO.download(); CDh = None if CD is None else CD.download()
        for k in range(T):
            if mode == 0: a, bb = acts[k, 0], acts[k, 1]
            elif mode == 1: a, bb = o.sample_actions_mixed(cur)
            elif mode == 2: a, bb = o.sample_actions_mixed(cur, mix[0], mix[1])
            else: a, bb = acts[k, 0], pol[cur]
            c = o.step(a, bb)
            assert np.array_equal(Oh[k], c["obs"]) and np.array_equal(Rh[k], c["reward"]), "rollout mode %d step %d %s" % (mode, k, tag)
            assert np.array_equal(TEh[k], c["terminated"]) and np.array_equal(TRh[k], c["truncated"]), "rollout flags " + tag
            assert FOh is None or np.array_equal(FOh[k], c["final_obs"]), "rollout final_obs mode %d step %d %s" % (mode, k, tag)
            assert CDh is None or np.array_equal(CDh[k], c["prob_code"]), "rollout prob_code mode %d step %d %s" % (mode, k, tag)
            cur = c["obs"]; lanes_steps += n
        s = b.get_state()
        for key, v in (("row_a", o.row_a), ("col_a", o.col_a), ("row_b", o.row_b), ("col_b", o.col_b), ("poss", o.poss & 1), ("needs_reset", (o.poss >> 1) & 1), ("t", o.t)):
            assert np.array_equal(s[key], v), "state %s %s" % (key, tag)
        assert b.tick == o.tick, "tick " + tag
    finally:
        b.close()
    rounds += 1
    if time.time() - last_note > 60:            # a progress line a minute (a silent GPU job is taken to be hung)
        last_note = time.time()
        print("  %.0f s: %d configurations, %.3g lane-steps" % (last_note - t0, rounds, lanes_steps), flush=True)
print("soak ok: %d random configurations, %.3g lane-steps compared bit for bit in %.0f s" % (rounds, lanes_steps, time.time() - t0))
