"""SoccerSimultaneousEnv — single-environment facade with the reference's Python surface.

Drop-in for `gym_soccer.envs.SoccerSimultaneousEnv`
(gym_soccer/envs/soccer_simultaneous_env.py:5-497): same constructor arguments (:35), same
`reset(seed, options)` (:410-424) / `step(action_dict)` (:375-408) signatures, dict-of-agents I/O,
scalar Python types, `AssertionError` on misuse, and the `env.state = tuple` injection hook the
reference's tests rely on.  The transition itself is NOT computed here: every step is one launch of
the HIP kernel on a 1-lane handle (libsoccer_hip.so, soccer_step_scalar: tuple, actions and uniform go in
as kernel arguments, the result comes back through a host-mapped record the call polls).

Randomness: like the reference this env owns an `np.random.RandomState` (MT19937, :57-58) and
draws exactly one uniform per reset and one per step (:395, :414); the uniform is handed to the
kernel, so for a given seed the trajectories are the reference's own, step for step.
"""
import ctypes as C

import numpy as np

from .. import _lib, spaces
from ..core import SoccerBatch


_ACTIONS = ['NOOP', 'NORTH', 'SOUTH', 'EAST', 'WEST']
_MOVES = {0: (0, 0), 1: (0, -1), 2: (0, 1), 3: (1, 0), 4: (-1, 0)}    # (dcol, drow) (:24-30)


class SoccerSimultaneousEnv:
    NOOP, NORTH, SOUTH, EAST, WEST = 0, 1, 2, 3, 4
    ACTION_STRING = _ACTIONS
    ACTION_STRING_TO_INT = {k: v for v, k in enumerate(_ACTIONS)}
    ACTION_INT_TO_MOVE = _MOVES
    ACTION_STRING_TO_MOVE = {_ACTIONS[k]: v for k, v in _MOVES.items()}
    MOVE_TO_ACTION_INT = {v: k for k, v in _MOVES.items()}
    MOVE_TO_ACTION_STRING = {v: _ACTIONS[k] for k, v in _MOVES.items()}
    TERMINAL_STATE = (-1, -1, -1, -1, -1)
    metadata = {"render_modes": ["ansi"]}

    def __init__(self, width=5, height=4, slip_prob=0.0, player_a_policy=None, player_b_policy=None,
                 seed=0, device=0):
        assert not (player_a_policy is not None and player_b_policy is not None), \
            "Both players cannot have a policy. At least one must be None."
        assert width >= 5, "Width must be at least 5 columns."
        assert height >= 4, "Height must be at least 4 rows."
        # one lane; a step is one kernel launch whose inputs travel as kernel arguments and whose result is
        # polled from a host-mapped record: no copies, no stream synchronisation
        self._batch = SoccerBatch(1, width, height, slip_prob, seed=seed, autoreset=False, device=device)
        self._io = _lib.ScalarIO()
        self._io_ref = C.byref(self._io)
        self._h = self._batch.h
        self._step_scalar = self._batch.lib.soccer_step_scalar
        self._reset_scalar = self._batch.lib.soccer_reset_scalar
        self._p_rounded = [np.round(p, 2) for p in self._batch.prob_table]     # info['p'] (:405), per prob_code
        self._max_t = self._batch.max_steps
        # single-agent mode: the fixed side's policy lives on the device and is looked up by the kernel
        if player_a_policy is not None:
            self._batch.set_policy('player_a', player_a_policy)
        if player_b_policy is not None:
            self._batch.set_policy('player_b', player_b_policy)
        self.width = width + 2                      # +2 goal columns (:48)
        self.height = height
        self.slip_prob = slip_prob
        self.seed = seed
        self.player_a_policy = player_a_policy
        self.player_b_policy = player_b_policy
        self.multiagent = player_a_policy is None and player_b_policy is None
        self.return_agent = ['player_a', 'player_b'] if self.multiagent else \
            ['player_a'] if player_a_policy is None else ['player_b']
        self.np_random = np.random.RandomState()
        self.np_random.seed(self.seed)
        self.goal_rows = ((height - 1) // 2, height // 2) if height % 2 == 0 else \
            (height // 2 - 1, height // 2, height // 2 + 1)
        self.goal_cols = (0, self.width - 1)

        # state_space / goal_states / unreachable_states from the library's rule tables
        lut, goal_value, isd = self._batch.tables()
        H, W = self.height, self.width
        self.unreachable_states, self.goal_states = [], {}
        self.state_space = {self.TERMINAL_STATE: 0}
        idx = np.arange(lut.size)
        p = idx & 1; r = idx >> 1
        yb = r % W; r //= W; xb = r % H; r //= H; ya = r % W; xa = r // W
        tuples = list(zip(xa.tolist(), ya.tolist(), xb.tolist(), yb.tolist(), p.tolist()))
        for f, st in enumerate(tuples):
            if lut[f] == 0xFFFF:
                self.unreachable_states.append(st)
            elif goal_value[f] != 0:
                self.goal_states[st] = float(goal_value[f])
            else:
                self.state_space[st] = int(lut[f])
        self.nS = self._batch.nS
        assert self.nS == len(self.state_space), "State space should be the same length as the number of states"
        self._reverse_state_space = {v: k for k, v in self.state_space.items()}
        self.nA = len(self.ACTION_STRING)
        self.observation_space = spaces.Dict({a: spaces.Discrete(self.nS) for a in self.return_agent})
        self.action_space = spaces.Dict({a: spaces.Discrete(self.nA) for a in self.return_agent})
        self.isd = [(1.0 / len(isd), tuple(int(x) for x in s)) for s in isd]
        self.needs_reset = True
        self.state = None                          # host mirror; authoritative (tests assign to it, :43)
        self.observations = None
        self.timestep = 0
        self.lastaction = None

    # ---- the reference's tables (P, P_readable, Pmat, Rmat; :167-293), built on first use ------------
    # The transition relation is enumerated ON THE DEVICE by the rule functions of the step kernels
    # (soccer_enumerate_transitions) in ~ms instead of the reference's 1 s (5x4) / 120 s (11x7) Python
    # loops; this method only arranges it into the reference's dict / tensor shapes, quirks included
    # (P[0] ends as the last goal tuple's lists, Pmat[0, 0] accumulates one unit per goal tuple).
    def _build_tables(self):
        count, prob, nxt, rew, done = self._batch.transitions()
        lut = self._batch.tables()[0]
        H, W = self.height, self.width
        idx = np.arange(lut.size)
        pp = idx & 1; r = idx >> 1
        yb = r % W; r //= W; xb = r % H; r //= H; ya = r % W; xa = r // W
        tuples = list(zip(xa.tolist(), ya.tolist(), xb.tolist(), yb.tolist(), pp.tolist()))
        obs_of = np.where(lut == 0xFFFF, 0, lut).astype(np.int64)
        flip = (not self.multiagent) and self.return_agent == ['player_b']      # :243-244
        P, P_readable = {}, {}
        nS, nA = self.nS, self.nA
        if self.multiagent:
            Pmat = np.zeros([nS, nS, nA, nA]); Rmat = np.zeros([nS, nA, nA])
        else:
            Pmat = np.zeros([nS, nS, nA]); Rmat = np.zeros([nS, nA])
        AS = self.ACTION_STRING
        for f in np.flatnonzero(count[:, 0] >= 0).tolist():
            st = tuples[f]; s_ = int(obs_of[f])
            P[s_] = {}; P_readable[st] = {}
            aaa = range(nA) if self.player_a_policy is None else [int(self.player_a_policy[s_])]
            aab = range(nA) if self.player_b_policy is None else [int(self.player_b_policy[s_])]
            for aa in aaa:
                for ab in aab:
                    ja = aa * 5 + ab
                    n = int(count[f, ja])
                    ps = prob[f, ja, :n].tolist(); ns = nxt[f, ja, :n].tolist()
                    rs = rew[f, ja, :n].tolist(); ds = done[f, ja, :n].tolist()
                    trs, trs_r = [], []
                    for k in range(n):
                        rr = float(rs[k])
                        if flip:
                            rr = -1 * rr
                        trs.append((ps[k], int(obs_of[ns[k]]), rr, bool(ds[k])))
                        trs_r.append((ps[k], tuples[ns[k]], rr, bool(ds[k])))
                    if self.multiagent:
                        P[s_][(aa, ab)] = trs; P_readable[st][(AS[aa], AS[ab])] = trs_r
                        Rmat[s_][aa][ab] = 0
                        for pr, nso, rr, _d in trs:
                            Pmat[s_][nso][aa][ab] += pr; Rmat[s_][aa][ab] += pr * rr
                    else:
                        a = aa if self.player_a_policy is None else ab
                        P[s_][a] = trs; P_readable[st][AS[a]] = trs_r
                        Rmat[s_][a] = 0
                        for pr, nso, rr, _d in trs:
                            Pmat[s_][nso][a] += pr; Rmat[s_][a] += pr * rr
        self._tables = (P, P_readable, Pmat, Rmat)

    def _table(self, i):
        if getattr(self, "_tables", None) is None:
            self._build_tables()
        return self._tables[i]

    P = property(lambda self: self._table(0))
    P_readable = property(lambda self: self._table(1))
    Pmat = property(lambda self: self._table(2))
    Rmat = property(lambda self: self._table(3))

    def _state_to_observation(self, state):
        state = self.TERMINAL_STATE if state in self.goal_states else state
        return self.state_space[state]

    def _observation_to_state(self, observation):
        return self._reverse_state_space[observation]

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.np_random.seed(seed)
        io = self._io
        io.u_reset = self.np_random.random()         # one uniform per reset (:414)
        rc = self._reset_scalar(self._h, self._io_ref)
        if rc:
            _lib.check(self._batch.lib, self._h, rc)
        self.state = (io.row_a, io.col_a, io.row_b, io.col_b, io.poss)
        p = np.round(1.0 / len(self.isd), 2)
        obs = io.obs
        self.observations = {a: obs for a in self.return_agent}
        infos = {a: {"p": p} for a in self.return_agent}
        self.lastaction = None
        self.needs_reset = False
        self.timestep = 0
        return self.observations, infos

    def step(self, action):
        assert not self.needs_reset, "Please reset the environment before taking a step"
        assert isinstance(action, dict), "Action must be a dictionary"
        assert len(action) == 1 or len(action) == 2, "Action must be a dictionary of length 1 or 2"
        assert self.player_a_policy is not None or 'player_a' in action, "A policy for player_a must be provided"
        assert self.player_b_policy is not None or 'player_b' in action, "A policy for player_b must be provided"
        io = self._io
        if self.multiagent:
            assert len(action) == 2, "Action must be a dictionary of length 2 for multiagent case"
            assert 'player_a' in action and 'player_b' in action, "Action must contain both 'player_a' and 'player_b'"
            aa, ab = int(action['player_a']), int(action['player_b'])
        else:
            assert len(action) == 1, "Action must be a dictionary of length 1 for single agent case"
            assert 'player_a' in action or 'player_b' in action, "Action must contain either 'player_a' or 'player_b'"
            # the fixed side's action is looked up by the kernel from the current observation (:187-188)
            aa = 0 if self.player_a_policy is not None else int(action['player_a'])
            ab = 0 if self.player_b_policy is not None else int(action['player_b'])
        assert 0 <= aa < 5 and 0 <= ab < 5, "actions must be in 0..4"
        # `env.state = tuple` (and env.timestep) set by the caller is what the step starts from
        # (tests/test_deterministic_soccer_simultaneous_env.py:43): the host mirror travels with the call
        st = self.state
        if st not in self.state_space and st not in self.goal_states or st == self.TERMINAL_STATE:
            raise KeyError(st)                       # P_readable[self.state] in the reference (:394)
        io.row_a, io.col_a, io.row_b, io.col_b, io.poss = st
        t = self.timestep
        io.t = 0 if t < 0 else (t if t < self._max_t else self._max_t)
        io.needs_reset = 0
        io.act_a = aa; io.act_b = ab
        io.u_step = self.np_random.random()          # one uniform per step (:395)
        rc = self._step_scalar(self._h, self._io_ref)
        if rc:
            _lib.check(self._batch.lib, self._h, rc)
        self.state = (io.row_a, io.col_a, io.row_b, io.col_b, io.poss)
        reward = float(io.reward)
        done = bool(io.terminated)
        obs = io.obs
        self.lastaction = action
        self.timestep = t = t + 1
        trunc = t >= 100
        p = self._p_rounded[io.prob_code]
        if self.multiagent:
            self.observations = {'player_a': obs, 'player_b': obs}
            rewards = {'player_a': reward, 'player_b': reward * -1}
            dones = {'player_a': done, 'player_b': done}
            truncateds = {'player_a': trunc, 'player_b': trunc}
            infos = {'player_a': {"p": p}, 'player_b': {"p": p}}
        else:                                        # learner-B tables store the flipped sign (:243-244)
            a = self.return_agent[0]
            self.observations = {a: obs}
            rewards = {a: reward if a == 'player_a' else -1 * reward}
            dones = {a: done}; truncateds = {a: trunc}; infos = {a: {"p": p}}
        self.needs_reset = done or trunc
        return self.observations, rewards, dones, truncateds, infos

    def render(self):
        """ASCII pitch (host-side debug print; the reference's :426-485)."""
        xa, ya, xb, yb, p = self.state
        W, H = self.width, self.height
        print(self.state)
        print(f"Player A position: x={xa}, y={ya}, possession={p == 0}")
        print(f"Player B position: x={xb}, y={yb}, possession={p == 1}")
        lines = ['  ' + '-' * (W * 2 - 4)]
        for r in range(H):
            cells = []
            for c in range(W):
                mark = '  '
                if (r, c) == (xa, ya): mark = 'A' + ('*' if p == 0 else ' ')
                if (r, c) == (xb, yb): mark = 'B' + ('*' if p == 1 else ' ')
                cells.append(mark)
            if r in self.goal_rows:
                left = cells[0] if '*' in cells[0] else '||'
                right = cells[-1] if '*' in cells[-1] else '||'
            else:
                left, right = ' |', '| '
            lines.append(left + ''.join(cells[1:-1]) + right)
        lines.append('  ' + '-' * (W * 2 - 4))
        for ln in lines:
            print(ln)
        print(f"Ball possession: {'A' if p == 0 else 'B'}")
        if self.lastaction:                                              # :462-473
            if self.multiagent:
                print("Last actions: A: %s, B: %s" % (self.ACTION_STRING[self.lastaction['player_a']],
                                                      self.ACTION_STRING[self.lastaction['player_b']]))
            else:
                who = self.return_agent[0]
                print("Last action: %s: %s" % ('A' if who == 'player_a' else 'B', self.ACTION_STRING[self.lastaction[who]]))
        carrier_col, carrier_row = (ya, xa) if p == 0 else (yb, xb)
        if carrier_row in self.goal_rows and carrier_col in self.goal_cols:
            scorer_is_a = carrier_col == W - 1
            own = (p == 0) != scorer_is_a
            who = 'A' if p == 0 else 'B'
            print(f"OWN GOAL! Player {who} scored in their own goal!" if own else f"GOAL! Player {who} scored!")

    def close(self):
        self._batch.close()
