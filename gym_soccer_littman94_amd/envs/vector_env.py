"""VectorSoccerEnv — N independent SoccerSimultaneous environments resident on one MI355X.

The reference has no vector env; this is the batched form of `SoccerSimultaneousEnv.step/reset`
(gym_soccer/envs/soccer_simultaneous_env.py:375-424) with the gym-0.26 `VectorEnv` conventions the
reference leaves undefined:
  * dict-of-agents I/O like the reference, every value an array of length `num_envs`;
  * auto-reset (default): a lane that terminates or truncates is reset inside the same step; the
    returned observation is the first one of the new episode and `infos["final_observation"]`
    carries the terminal one (`infos["_final_observation"]` marks the lanes it applies to);
  * rewards float32 (+r for player_a, -r for player_b, :400-402), terminated/truncated bool.

Two I/O modes:
  * numpy (default): actions are host arrays; results are numpy views over the handle's pinned
    staging block when `copy=False` (one copy in, one kernel, one copy out over PCIe per step; valid
    until the next step()), private copies with the default `copy=True` (gym.vector's convention);
  * device: actions are torch CUDA int8 tensors; results are torch tensors living in buffers the env
    owns, nothing is synchronised — the mode for rollout loops that stay on the GPU.  The float32 rewards of the
    returned agents and `infos["_final_observation"]` are written by the step kernel itself (`float_rewards=True`, the
    default: +4 B per env-step and agent instead of a 5 us cast kernel per read); `float_rewards=False` leaves them to
    be computed on first access and `reward_int8` is then the zero-cost way to read player A's reward.
    `info=False` is the lean form for loops that read observations, rewards and the done flags only: no `info[agent]["p"]`,
    no `infos["final_observation"]`, no episode histogram, no int8 reward stream, and ONE float32 reward stream (player_a's
    when both agents are returned; player_b's is its negation, computed on first access) — 23 instead of 31 bytes per
    env-step and the step kernel's instantiation without the second observation index
    (`infos["_final_observation"]`, the lanes whose episode just ended, stays).
Per-lane randomness is Philox4x32-10 keyed by (seed, global lane id, tick): include/soccer_hip.h.
"""
import numpy as np

from .. import spaces
from ..core import SoccerBatch

AGENTS = ('player_a', 'player_b')


class _Lazy(dict):
    """A dict whose values are computed on first access (and cached until `invalidate()`): device mode returns one
    of these for everything that would cost an extra kernel launch per step and that the step kernel does not write itself — `info[agent]["p"]` (a gather + a
    cast), the float32 rewards (a cast; a negation for player_b) and `infos["_final_observation"]` (an OR) — so a
    rollout loop that does not look at them pays one launch per step and nothing else."""
    def __init__(self, thunks, eager=None, dirty=None):
        super().__init__(eager or {})
        self._thunks = thunks
        self._eager = dict(eager or {})
        self._dirty = dirty            # the owner's list of dicts holding a computed value (what the next step must drop)

    def invalidate(self):
        dict.clear(self)
        dict.update(self, self._eager)

    def __missing__(self, key):
        if key not in self._thunks:
            raise KeyError(key)
        v = self._thunks[key]()
        self[key] = v
        if self._dirty is not None:
            self._dirty.append(self)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __contains__(self, key):
        return key in self._thunks or dict.__contains__(self, key)

    def keys(self):
        return list(dict.fromkeys(list(self._eager) + list(self._thunks)))

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class _LazyInfo(_Lazy):
    """info[agent]: 'p' computed on first access."""
    def __init__(self, make_p, dirty=None):
        super().__init__({"p": make_p}, dirty=dirty)


class VectorSoccerEnv:
    metadata = {"render_modes": []}
    # what code written against gym 0.26's `gym.vector.VectorEnv` looks for besides reset / step / the four spaces
    is_vector_env = True
    render_mode = None
    spec = None
    closed = False

    def __init__(self, num_envs, width=5, height=4, slip_prob=0.0, seed=0, autoreset=True,
                 device=0, lane_offset=0, max_episode_steps=100, io="numpy", strict=True,
                 envs_per_thread=0, player_a_policy=None, player_b_policy=None, copy=True, float_rewards=True, info=True,
                 stream_actions=False):
        assert io in ("numpy", "device"), "io must be 'numpy' or 'device'"
        assert not (player_a_policy is not None and player_b_policy is not None), \
            "Both players cannot have a policy. At least one must be None."
        self.num_envs = int(num_envs)
        self.io = io
        self.strict = strict
        self.float_rewards = bool(float_rewards)
        self.info = bool(info) or io != "device"    # the lean form exists for device io only
        self.copy = bool(copy)      # numpy io: return copies (gym.vector's default) or views over the staging block
        stream = None
        if io == "device":
            import torch
            self._torch = torch
            self._dev = torch.device("cuda", device)
            stream = torch.cuda.current_stream(self._dev).cuda_stream   # results are ordered with torch work
        self._batch = SoccerBatch(self.num_envs, width, height, slip_prob, seed=seed, autoreset=autoreset,
                                  max_steps=max_episode_steps, device=device, lane_offset=lane_offset,
                                  stream=stream, envs_per_thread=envs_per_thread, step_stats=self.info,
                                  stream_actions=stream_actions)    # (SOCCER_F_STREAM_ACTIONS: see SoccerBatch)
        b = self._batch
        self.width, self.height, self.slip_prob = width + 2, height, slip_prob
        self.nS, self.nA = b.nS, b.nA
        self.autoreset = bool(autoreset)
        # single-agent mode (reference :54-56): the side with a policy is played by the kernel
        self.player_a_policy, self.player_b_policy = player_a_policy, player_b_policy
        self.multiagent = player_a_policy is None and player_b_policy is None
        self.return_agent = list(AGENTS) if self.multiagent else \
            ['player_a'] if player_a_policy is None else ['player_b']
        if player_a_policy is not None:
            b.set_policy('player_a', player_a_policy)
        if player_b_policy is not None:
            b.set_policy('player_b', player_b_policy)
        ags = self.return_agent
        self.single_observation_space = spaces.Dict({a: spaces.Discrete(self.nS) for a in ags})
        self.single_action_space = spaces.Dict({a: spaces.Discrete(self.nA) for a in ags})
        self.observation_space = spaces.Dict(
            {a: spaces.MultiDiscrete(np.full(self.num_envs, self.nS)) for a in ags})
        self.action_space = spaces.Dict(
            {a: spaces.MultiDiscrete(np.full(self.num_envs, self.nA)) for a in ags})
        self._needs_reset = True
        self._steps = 0
        self._p_rounded = np.round(b.prob_table, 2)
        if io == "device":
            t, n, d = self._torch, self.num_envs, self._dev
            u16 = getattr(t, "uint16", t.int16)
            self._obs = t.zeros(n, dtype=u16, device=d); self._fin = t.zeros(n, dtype=u16, device=d)
            lean = not self.info
            self._rew = None if (lean and self.float_rewards) else t.zeros(n, dtype=t.int8, device=d)
            self._term = t.zeros(n, dtype=t.uint8, device=d); self._trunc = t.zeros(n, dtype=t.uint8, device=d)
            self._code = None if lean else t.zeros(n, dtype=t.uint8, device=d)
            if lean: self._fin = None
            self._prob = t.tensor(np.round(b.prob_table, 2), dtype=t.float64, device=d)
            # Everything step() returns is built ONCE: the result buffers are fixed, so the dicts of views over them
            # are too (the reference also hands back the same dict objects every call); what needs a kernel of its
            # own is computed on first access.  step() itself = one ctypes call = one kernel launch.
            ags = self.return_agent
            term_b, trunc_b = self._term.view(t.bool), self._trunc.view(t.bool)      # 0/1 bytes: reinterpret, no kernel
            self._ret_obs = {ag: self._obs for ag in ags}
            self._ret_term = {ag: term_b for ag in ags}
            self._ret_trunc = {ag: trunc_b for ag in ags}
            self._stale = []            # lazy dicts that computed something since the last step
            f32 = {}
            if self.float_rewards:      # the kernel writes the floats (and terminated | truncated) next to the int8 stream
                # lean form: the game is zero-sum (:400-402), so ONE float stream is written — the first returned agent's —
                # and the other agent's reward is its negation, computed on first access
                written = ags[:1] if lean else ags
                f32 = {ag: t.zeros(n, dtype=t.float32, device=d) for ag in written}
                self._finished = t.zeros(n, dtype=t.uint8, device=d)
                if len(written) == len(ags):
                    self._ret_rew = dict(f32)
                else:
                    first = f32[ags[0]]
                    self._ret_rew = _Lazy({ags[1]: lambda: 0.0 - first}, dict(f32), dirty=self._stale)
            else:
                rew_thunks = {}
                if 'player_a' in ags: rew_thunks['player_a'] = lambda: self._rew.to(t.float32)
                if 'player_b' in ags: rew_thunks['player_b'] = lambda: 0.0 - self._rew.to(t.float32)      # :400-402, :243-244
                self._ret_rew = _Lazy(rew_thunks, dirty=self._stale)
            eager = {}
            if not lean:
                self._ret_p = _LazyInfo(lambda: self._prob[self._code.long()], dirty=self._stale)       # np.round(prob, 2) of the sampled transition (:405)
                eager = {ag: self._ret_p for ag in ags}
                eager["final_observation"] = {ag: self._fin for ag in ags}
            if self.float_rewards:
                eager["_final_observation"] = self._finished.view(t.bool)
                self._ret_infos = eager
            else:
                self._ret_infos = _Lazy({"_final_observation": lambda: term_b | trunc_b}, eager, dirty=self._stale)
            from .._lib import StepArgs
            fptr = lambda ag: f32[ag].data_ptr() if ag in f32 else None
            dptr = lambda x: None if x is None else x.data_ptr()
            self._step_args = StepArgs(None, None, None, None, self._obs.data_ptr(), dptr(self._rew),
                                       self._term.data_ptr(), self._trunc.data_ptr(), dptr(self._code),
                                       dptr(self._fin), None, fptr('player_a'), fptr('player_b'),
                                       self._finished.data_ptr() if self.float_rewards else None)
            self._step_call = b.lib.batched_step_ex
            import ctypes
            self._step_ref = ctypes.byref(self._step_args)

    # ---------------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None, mask=None):
        """Reset every lane (or the lanes selected by `mask`, e.g. the finished ones when auto-reset
        is off).  `seed` re-keys the per-lane Philox streams (np_random.seed in the reference, :411-412)."""
        b = self._batch
        if self.strict and self.io == "device" and b.peek_misuse():     # what the steps before this reset raised (strict mode
            self._raise_on_misuse()                                       # looks every 4th step only: nothing is lost here)
        if seed is not None:
            b.seed(seed)
        p = np.round(1.0 / b.n_isd, 2)
        if self.io == "numpy":
            obs = b.reset_host(mask=mask)
            infos = {a: {"p": np.full(self.num_envs, p)} for a in self.return_agent}
            self._needs_reset = False
            return {a: obs for a in self.return_agent}, infos
        m = None if mask is None else mask.to(self._torch.uint8)
        b.reset(mask=m, obs=self._obs)
        self._needs_reset = False
        infos = {a: {"p": self._torch.full((self.num_envs,), float(p), device=self._dev)} for a in self.return_agent}
        return {a: self._obs for a in self.return_agent}, infos

    def _check_actions(self, action):
        assert isinstance(action, dict), "Action must be a dictionary"
        if self.multiagent:
            assert len(action) == 2, "Action must be a dictionary of length 2 for multiagent case"
            assert 'player_a' in action and 'player_b' in action, "Action must contain both 'player_a' and 'player_b'"
        else:
            assert len(action) == 1, "Action must be a dictionary of length 1 for single agent case"
            assert self.return_agent[0] in action, "Action must contain the learner's key only"
        return action.get('player_a'), action.get('player_b')

    def _rewards(self, r):
        """player A's reward array -> dict per returned agent; B's is the negation (:400-402, :243-244)."""
        out = {}
        if 'player_a' in self.return_agent: out['player_a'] = r
        if 'player_b' in self.return_agent: out['player_b'] = -r
        return out

    def step(self, action):
        assert not self._needs_reset, "Please reset the environment before taking a step"
        a, bb = self._check_actions(action)
        b = self._batch
        n = self.num_envs
        if self.io == "numpy":
            # actions are written straight into the handle's pinned staging block and the results are
            # views over it (valid until the next step): nothing is copied on the host side
            stg = b.staging()
            for key, x in (("act_a", a), ("act_b", bb)):
                if x is None:
                    continue
                x = np.asarray(x)
                assert x.shape == (n,), "one action per environment and agent"
                if x.dtype == np.int8:                      # one reduction: negative values read as >= 128
                    assert n == 0 or x.view(np.uint8).max() < self.nA, "actions must be in 0..4"
                else:
                    assert ((x >= 0) & (x < self.nA)).all(), "actions must be in 0..4"
                np.copyto(stg[key], x, casting="unsafe")
            b.step_staged(act_a=a is not None, act_b=bb is not None)
            out = {k: stg[k].copy() for k in ("obs", "final_obs", "reward", "terminated", "truncated", "prob_code")} \
                if self.copy else stg
            if self.strict:
                self._raise_on_misuse()
            r = out["reward"].astype(np.float32)
            term = out["terminated"].view(np.bool_); trunc = out["truncated"].view(np.bool_)
            fin_mask = term | trunc
            ags = self.return_agent
            code = out["prob_code"]
            lazy = _LazyInfo(lambda: self._p_rounded[code])     # np.round(prob, 2) of the sampled transition (:405)
            infos = {ag: lazy for ag in ags}
            infos["final_observation"] = {ag: out["final_obs"] for ag in ags}
            infos["_final_observation"] = fin_mask
            return ({ag: out["obs"] for ag in ags}, self._rewards(r),
                    {ag: term for ag in ags}, {ag: trunc for ag in ags}, infos)
        i8 = self._torch.int8
        for x in (a, bb):
            assert x is None or (x.dtype is i8 and x.is_cuda and x.ndim == 1 and x.shape[0] == n and x.is_contiguous()), \
                "device io expects contiguous torch.int8 CUDA tensors"
        # Action VALUES cannot be asserted here without a device round trip.  The kernels execute a byte b as the move
        # table[b & 7] with 5..7 = NOOP (never an out-of-table access) and raise a sticky flag for any byte outside 0..4;
        # strict mode looks at the flags of the launches completed so far (a host-mapped word, no synchronisation; every
        # fourth step — this call is host-bound at 2^20 lanes, every microsecond of Python shows), so a bad action or a step
        # on finished lanes surfaces as the reference's AssertionError up to 4 steps late; reset(), rollout(),
        # episode_histogram() and close() look too, so every misuse is reported eventually.
        self._steps = steps = self._steps + 1
        if self.strict and not (steps & 3) and b.peek_misuse():
            self._raise_on_misuse()
        args = self._step_args
        args.act_a = a.data_ptr() if a is not None else None
        args.act_b = bb.data_ptr() if bb is not None else None
        code = self._step_call(b.h, self._step_ref)
        if code:
            b._check(code)
        if self._stale:                                  # only what the caller actually looked at is dropped
            for lz in self._stale:
                lz.invalidate()
            del self._stale[:]
        return self._ret_obs, self._ret_rew, self._ret_term, self._ret_trunc, self._ret_infos

    # ---------------------------------------------------------------------------------------------
    def rollout(self, n_steps, actions=None, sample_actions=False, mixed_policies=None, infos="last"):
        """T = `n_steps` fused steps — by definition EXACTLY what T successive `step()` calls return, stacked over T (same ticks,
        same auto-reset convention: soccer_simultaneous_env.py:397-408 per step) — with the state in registers for the whole run.

        actions  dict like step()'s, every value [T, num_envs] (device io: contiguous torch.int8 CUDA tensors; numpy io: integer
                 arrays, checked to be 0..4 on the host).  Single-agent mode: the learner's key only.
        sample_actions=True (actions None)  both players act uniformly at random, drawn in the kernel from the lanes' purpose-1
                 Philox words (include/soccer_hip.h); `mixed_policies` = {agent: [nS, 5] probabilities} samples that agent's
                 action from its row of the current observation instead (BASELINE config 5).
        infos    which of step()'s infos come back:
                 "last" (default)  the LAST step's, exactly as after the T-th step() call — `batched_rollout` for T - 1 steps and
                         one full `batched_step_ex` (7 B per env-step; with sampled actions all T steps are fused and only
                         `_final_observation` of the last step is reported);
                 "all"   every step's, stacked over T like everything else: infos["final_observation"][agent] [T, num_envs] (the
                         observation before that step's auto-reset — what a learner bootstraps from at a truncation),
                         infos["_final_observation"] [T, num_envs], infos[agent]["p"] [T, num_envs] — ONE `batched_rollout_ex`
                         launch (10 B per env-step and a second observation index per step);
                 "none"  neither (what an `info=False` env gives whatever is asked): `_final_observation` [T, num_envs] on access.

        Returns (observations, rewards, terminated, truncated, infos): dicts per returned agent of [T, num_envs] arrays —
        uint16 observations, float32 rewards (device io: cast from the kernel's int8 trajectory on first access; the int8
        trajectory of player A's reward is infos["reward_int8"]), bool flags.  The arrays are buffers the env owns, one set per
        (T, infos), overwritten by the next such rollout."""
        assert not self._needs_reset, "Please reset the environment before taking a step"
        assert infos in ("last", "all", "none"), "infos must be 'last', 'all' or 'none'"
        T, n, b = int(n_steps), self.num_envs, self._batch
        assert T >= 1, "n_steps must be >= 1"
        mode = infos if self.info else "none"
        ags = self.return_agent
        if sample_actions:
            assert actions is None and self.multiagent, "sample_actions: no action streams, both players are sampled"
            a = bb = None
        else:
            assert mixed_policies is None, "mixed_policies needs sample_actions=True"
            a, bb = self._check_actions(actions)
        if self.io == "device":
            return self._rollout_device(T, a, bb, sample_actions, mixed_policies, mode)
        # ---- numpy io: host arrays in, host arrays out; the same launches, one upload and one download ------------------
        dev = {}
        for key, x in (("a", a), ("b", bb)):
            if x is None:
                continue
            x = np.asarray(x)
            assert x.shape == (T, n), "one action per step, environment and agent: [n_steps, num_envs]"
            assert ((x >= 0) & (x < self.nA)).all(), "actions must be in 0..4"
            dev[key] = b.alloc((T, n), np.int8).upload(x.astype(np.int8, copy=False))
        obs = b.alloc((T, n), np.uint16); rew = b.alloc((T, n), np.int8)
        term = b.alloc((T, n), np.uint8); trunc = b.alloc((T, n), np.uint8)
        last_only = mode == "last" and not sample_actions
        rows = 1 if last_only else T
        fin = b.alloc((rows, n), np.uint16) if mode != "none" else None
        code = b.alloc((rows, n), np.uint8) if mode != "none" else None
        mix = self._mix_tables(mixed_policies, lambda t: b.alloc(t.shape, np.uint16).upload(t)) if sample_actions else {}
        kw = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, out_stride=n)
        if last_only:
            if T > 1:
                b.rollout(T - 1, dev.get("a"), dev.get("b"), act_stride=n, **kw)
            k = T - 1
            row = lambda arr: None if arr is None else arr.row(k)
            b.step(row(dev.get("a")), row(dev.get("b")), obs=obs.row(k), reward=rew.row(k), terminated=term.row(k), truncated=trunc.row(k),
                   prob_code=code, final_obs=fin)
        else:
            full = mode == "all"
            b.rollout(T, dev.get("a"), dev.get("b"), act_stride=n, sample_actions=sample_actions, mix_a=mix.get("player_a"),
                      mix_b=mix.get("player_b"), final_obs=fin if full else None, prob_code=code if full else None, **kw)
        O, R8 = obs.download(), rew.download()
        TE, TR = term.download().view(np.bool_), trunc.download().view(np.bool_)
        out_infos = {"reward_int8": R8}
        if mode == "all":
            c, f = code.download(), fin.download()
            lazy = _LazyInfo(lambda: self._p_rounded[c])
            out_infos.update({ag: lazy for ag in ags}); out_infos["final_observation"] = {ag: f for ag in ags}
            out_infos["_final_observation"] = TE | TR
        elif last_only:
            c, f = code.download()[0], fin.download()[0]
            lazy = _LazyInfo(lambda: self._p_rounded[c])
            out_infos.update({ag: lazy for ag in ags}); out_infos["final_observation"] = {ag: f for ag in ags}
            out_infos["_final_observation"] = TE[-1] | TR[-1]
        else:
            out_infos["_final_observation"] = (TE[-1] | TR[-1]) if mode == "last" else (TE | TR)
        for x in list(dev.values()) + [obs, rew, term, trunc] + [y for y in (fin, code) if y is not None] + list(mix.values()):
            x.free()
        if self.strict:
            self._raise_on_misuse()
        return ({ag: O for ag in ags}, self._rewards(R8.astype(np.float32)), {ag: TE for ag in ags}, {ag: TR for ag in ags}, out_infos)

    def _mix_tables(self, mixed_policies, put):
        out = {}
        for ag, probs in (mixed_policies or {}).items():
            assert ag in AGENTS, "mixed_policies keys are 'player_a' / 'player_b'"
            out[ag] = put(SoccerBatch.mixed_policy_thresholds(probs))
        return out

    def _rollout_device(self, T, a, bb, sample, mixed_policies, mode):
        t, n, d, b = self._torch, self.num_envs, self._dev, self._batch
        for x in (a, bb):
            assert x is None or (x.dtype is t.int8 and x.is_cuda and x.shape == (T, n) and x.is_contiguous()), \
                "device io expects contiguous torch.int8 CUDA tensors of shape [n_steps, num_envs]"
        if self.strict and b.peek_misuse():           # (like step(): what the launches completed so far have raised)
            self._raise_on_misuse()
        key = (T, mode, bool(sample))
        plan = getattr(self, "_roll_plan", None)
        if plan is None or plan["key"] != key:
            plan = self._roll_plan = self._make_roll_plan(T, mode, bool(sample))     # one set of buffers: another shape replaces it
        for lz in plan["lazy"]:                       # what the caller looked at after the previous rollout
            lz.invalidate()
        ra = plan["rollout_args"]
        pa = a.data_ptr() if a is not None else None
        pb = bb.data_ptr() if bb is not None else None
        ra.act_a = pa; ra.act_b = pb
        if sample:
            mix = self._mix_tables(mixed_policies, lambda tab: t.from_numpy(tab.view(np.int16)).to(d))
            ra.mix_a = mix["player_a"].data_ptr() if "player_a" in mix else None
            ra.mix_b = mix["player_b"].data_ptr() if "player_b" in mix else None
            plan["mix_keep"] = mix                    # alive until the next rollout: the launch that reads them is asynchronous
        if plan["last_only"]:
            # T - 1 fused steps, then the last one as a full step: two ctypes calls on argument blocks built once
            if T > 1:
                code = plan["rollout_call"](b.h, plan["rollout_ref"])
                if code:
                    b._check(code)
            sa = plan["step_args"]; last = (T - 1) * n
            sa.act_a = None if pa is None else pa + last
            sa.act_b = None if pb is None else pb + last
            code = plan["step_call"](b.h, plan["step_ref"])
        else:
            code = plan["rollout_ex_call"](b.h, plan["rollout_ref"], plan["extra_ref"])     # ONE ctypes call = one launch for the T steps
        if code:
            b._check(code)
        return plan["ret"]

    def _make_roll_plan(self, T, mode, sample):
        import ctypes
        from .._lib import RolloutArgs, RolloutExtra, StepArgs
        t, n, d, b = self._torch, self.num_envs, self._dev, self._batch
        u16 = getattr(t, "uint16", t.int16)
        obs = t.empty((T, n), dtype=u16, device=d); rew = t.empty((T, n), dtype=t.int8, device=d)
        term = t.empty((T, n), dtype=t.uint8, device=d); trunc = t.empty((T, n), dtype=t.uint8, device=d)
        ags = self.return_agent
        term_b, trunc_b = term.view(t.bool), trunc.view(t.bool)
        thunks = {}
        if 'player_a' in ags: thunks['player_a'] = lambda: rew.to(t.float32)
        if 'player_b' in ags: thunks['player_b'] = lambda: 0.0 - rew.to(t.float32)          # :400-402, :243-244
        rewards = _Lazy(thunks)
        lazies = [rewards]
        eager = {"reward_int8": rew}
        last_only = mode == "last" and not sample
        fin = code = finished = None
        if mode == "all" or last_only:
            shape = (n,) if last_only else (T, n)
            fin = t.empty(shape, dtype=u16, device=d); code = t.empty(shape, dtype=t.uint8, device=d)
            p_lazy = _LazyInfo(lambda: self._prob[code.long()])                             # np.round(prob, 2) of the sampled transition(s) (:405)
            lazies.append(p_lazy)
            eager.update({ag: p_lazy for ag in ags})
            eager["final_observation"] = {ag: fin for ag in ags}
        if last_only:
            finished = t.empty(n, dtype=t.uint8, device=d)
            eager["_final_observation"] = finished.view(t.bool)
            infos = eager
        else:
            infos = _Lazy({"_final_observation": (lambda: term_b[-1] | trunc_b[-1]) if mode == "last" else (lambda: term_b | trunc_b)}, eager)
            lazies.append(infos)
        ret = ({ag: obs for ag in ags}, rewards, {ag: term_b for ag in ags}, {ag: trunc_b for ag in ags}, infos)
        rollout_args = RolloutArgs(max(T - 1, 1) if last_only else T, 1 if sample else 0, None, None, n, obs.data_ptr(), rew.data_ptr(),
                                   term.data_ptr(), trunc.data_ptr(), n, None, None, None, None)
        full = mode == "all"
        extra = RolloutExtra(fin.data_ptr() if full else None, code.data_ptr() if full else None)
        plan = {"key": (T, mode, sample), "bufs": (obs, rew, term, trunc, fin, code, finished), "lazy": tuple(lazies), "ret": ret,
                "last_only": last_only, "rollout_args": rollout_args, "rollout_ref": ctypes.byref(rollout_args), "extra": extra,
                "extra_ref": ctypes.byref(extra), "rollout_call": b.lib.batched_rollout, "rollout_ex_call": b.lib.batched_rollout_ex}
        if last_only:
            k = T - 1
            step_args = StepArgs(None, None, None, None, obs[k].data_ptr(), rew[k].data_ptr(), term[k].data_ptr(), trunc[k].data_ptr(),
                                 code.data_ptr(), fin.data_ptr(), None, None, None, finished.data_ptr())
            plan.update(step_args=step_args, step_ref=ctypes.byref(step_args), step_call=b.lib.batched_step_ex)
        return plan

    @property
    def reward_int8(self):
        """device io: player A's reward of the last step as the int8 tensor the kernel wrote (-1 / 0 / +1), no cast
        (None with info=False and float rewards: that stream is not written then)."""
        return self._rew

    def _raise_on_misuse(self):
        flags = self._batch.misuse()
        if flags:
            self._batch.reset_stats()
            if flags & SoccerBatch.MISUSE_ACTION:
                raise AssertionError("actions must be in 0..4 (an action byte outside that range reached the device; "
                                     "it was executed as a move inside the pitch)")
            raise AssertionError("Please reset the environment before taking a step "
                                 "(some lanes had terminated or truncated; they were left untouched)")

    # ---------------------------------------------------------------------------------------------
    def get_state(self):
        """dict of int8/uint8 arrays: row_a, col_a, row_b, col_b, poss, t, needs_reset."""
        return self._batch.get_state()

    def set_state(self, **kw):
        self._batch.set_state(**kw)
        self._needs_reset = False

    def checkpoint(self):
        """State streams + Philox (seed, tick): restoring it on any VectorSoccerEnv of the same shape (and
        lane_offset) reproduces every later step."""
        return self._batch.checkpoint()

    def restore(self, ck):
        self._batch.restore(ck)
        self._needs_reset = False

    def episode_histogram(self):
        """Counts of finished episodes by player A's return (-1, 0, +1); not collected with info=False.  Synchronises, so
        strict mode reports any misuse the steps so far have raised."""
        assert self.info, "the episode histogram is not collected with info=False"
        hist, flags = self._batch.stats()
        if self.strict and flags:
            self._raise_on_misuse()
        return hist

    @property
    def batch(self):
        return self._batch

    @property
    def unwrapped(self):
        return self

    # gym.vector.VectorEnv's asynchronous pair and its reset counterpart: the launch IS asynchronous with device io (nothing is
    # synchronised until a result is read); with numpy io the work happens in the *_wait call
    def step_async(self, actions):
        self._pending_actions = actions
        if self.io == "device":
            self._pending_result = self.step(actions)

    def step_wait(self, **kwargs):
        assert getattr(self, "_pending_actions", None) is not None, "step_wait() without step_async()"
        actions, self._pending_actions = self._pending_actions, None
        if self.io == "device":
            out, self._pending_result = self._pending_result, None
            return out
        return self.step(actions)

    def reset_async(self, seed=None, options=None):
        self._pending_reset = (seed, options)

    def reset_wait(self, seed=None, options=None, **kwargs):
        s, o = getattr(self, "_pending_reset", None) or (seed, options)
        self._pending_reset = None
        return self.reset(seed=s, options=o)

    def close_extras(self, **kwargs):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __repr__(self):
        return "VectorSoccerEnv(num_envs=%d, %dx%d, slip_prob=%g, io=%r)" % (self.num_envs, self.width - 2, self.height, self.slip_prob, self.io)

    def close(self):
        """Frees the handle.  strict mode: a misuse raised by the last steps (device io reports up to 4 steps late, see step())
        is reported here at the latest — after the handle is gone, as the reference's assert would have been (:376, :393)."""
        self.closed = True
        flags = 0
        if self.strict and self._batch.h:
            try:
                flags = self._batch.misuse()         # synchronises: everything enqueued has run
            except Exception:
                flags = 0
        self._batch.close()
        if flags & SoccerBatch.MISUSE_ACTION:
            raise AssertionError("actions must be in 0..4 (an action byte outside that range reached the device; "
                                 "it was executed as a move inside the pitch)")
        if flags:
            raise AssertionError("Please reset the environment before taking a step "
                                 "(some lanes had terminated or truncated; they were left untouched)")
