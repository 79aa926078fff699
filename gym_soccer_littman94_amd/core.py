"""SoccerBatch — object wrapper over one libsoccer_hip handle (N lanes resident on one MI355X).

This is the thin host layer between the gym-style classes (env.py, vector_env.py) and the C ABI.
It never computes a transition itself: every step/reset is a kernel launch in libsoccer_hip.so.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import Config, RolloutArgs, StepArgs


class DeviceArray:
    """A caller-owned device buffer allocated through the handle (no torch needed)."""

    def __init__(self, batch, shape, dtype):
        self.batch = batch
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        batch._check(batch.lib.soccer_malloc(batch.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host):
        a = np.ascontiguousarray(host, dtype=self.dtype)
        assert a.nbytes == self.nbytes, "size mismatch: %d vs %d bytes" % (a.nbytes, self.nbytes)
        b = self.batch
        b._check(b.lib.soccer_memcpy_h2d(b.h, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def upload_rows(self, first_row, host):
        """Copy `host` (rows of the same width) into rows first_row.. of a 2-D+ buffer."""
        a = np.ascontiguousarray(host, dtype=self.dtype)
        row_bytes = int(np.prod(self.shape[1:])) * self.dtype.itemsize
        assert a.nbytes % row_bytes == 0 and int(first_row) * row_bytes + a.nbytes <= self.nbytes
        b = self.batch
        b._check(b.lib.soccer_memcpy_h2d(b.h, self.ptr + int(first_row) * row_bytes, a.ctypes.data, a.nbytes))
        return self

    def download_rows(self, first_row, n_rows):
        row_bytes = int(np.prod(self.shape[1:])) * self.dtype.itemsize
        out = np.empty((int(n_rows),) + self.shape[1:], self.dtype)
        b = self.batch
        b._check(b.lib.soccer_memcpy_d2h(b.h, out.ctypes.data, self.ptr + int(first_row) * row_bytes, out.nbytes))
        return out

    def download(self, out=None):
        if out is None:
            out = np.empty(self.shape, self.dtype)
        b = self.batch
        b._check(b.lib.soccer_memcpy_d2h(b.h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def fill(self, value):
        b = self.batch
        b._check(b.lib.soccer_memset(b.h, self.ptr, int(value), self.nbytes))
        return self

    def row(self, k):
        """Device pointer of row k of a 2-D buffer."""
        return self.ptr + int(k) * self.shape[-1] * self.dtype.itemsize

    def free(self):
        if self.ptr and self.batch.h:
            self.batch.lib.soccer_free(self.batch.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    """Device pointer of a DeviceArray / torch tensor / int / None."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):          # torch tensor on the handle's device
        return x.data_ptr()
    raise TypeError("expected a DeviceArray, a device tensor or an integer address, got %r" % type(x))


class SoccerBatch:
    """N lanes of the Littman-94 soccer game resident on one GPU.

    Constructor arguments mirror SoccerSimultaneousEnv.__init__
    (gym_soccer/envs/soccer_simultaneous_env.py:35); failed validations raise AssertionError as the
    reference's asserts do (:45-46).
    """

    def __init__(self, n_lanes, width=5, height=4, slip_prob=0.0, seed=0, autoreset=False,
                 max_steps=100, device=0, lane_offset=0, stream=None, envs_per_thread=0, host_mapped=False, step_stats=True,
                 stream_actions=False):
        """step_stats: batched_step also feeds the episode histogram (~5 % of a launch; on by default here,
        off in the raw C ABI).
        stream_actions: batched_step reads its action streams with the non-temporal hint (SOCCER_F_STREAM_ACTIONS: action data
        that is walked through once and does not fit the Infinity Cache).
        stream: None -> the handle creates its own HIP stream; an integer hipStream_t -> enqueue on
        that stream (0 = the device's default/null stream, which is what torch's default stream is)."""
        self.lib = _lib.load()
        self.h = None
        self._arrays = weakref.WeakSet()        # device buffers handed out by alloc(); freed with the handle
        cfg = Config(n_lanes=int(n_lanes), width=int(width), height=int(height),
                     slip_prob=float(slip_prob), max_steps=int(max_steps), device=int(device),
                     seed=int(seed) & 0xFFFFFFFFFFFFFFFF, lane_offset=int(lane_offset),
                     flags=(_lib.F_AUTORESET if autoreset else 0) | (_lib.F_NULL_STREAM if stream == 0 else 0) |
                     (_lib.F_HOST_MAPPED if host_mapped else 0) | (_lib.F_STEP_STATS if step_stats else 0) |
                     (_lib.F_STREAM_ACTIONS if stream_actions else 0),
                     envs_per_thread=int(envs_per_thread), stream=stream or None)
        h = C.c_void_p()
        _lib.check(self.lib, None, self.lib.soccer_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.n = int(n_lanes)
        self.width, self.height, self.slip_prob = int(width), int(height), float(slip_prob)
        self.autoreset, self.max_steps = bool(autoreset), int(max_steps)
        self.device, self.lane_offset = int(device), int(lane_offset)
        ns, ll, ni, iw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.soccer_dims(self.h, C.byref(ns), C.byref(ll), C.byref(ni), C.byref(iw)))
        self.nS, self.lut_len, self.n_isd, self.internal_width = ns.value, ll.value, ni.value, iw.value
        self.nA = 5
        pt = (C.c_double * 12)()
        self._check(self.lib.soccer_prob_table(self.h, C.byref(pt)))
        self.prob_table = np.array(pt, dtype=np.float64)

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, code):
        _lib.check(self.lib, self.h, code)

    def close(self):
        if self.h:
            for a in list(self._arrays):            # buffers that outlived their users: no leak past the handle
                a.free()
            self.lib.soccer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        self._arrays.add(a)
        return a

    def sync(self):
        self._check(self.lib.soccer_sync(self.h))

    def seed(self, seed):
        self._check(self.lib.soccer_seed(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    @property
    def tick(self):
        return int(self.lib.soccer_tick(self.h))

    # -- checkpoint / resume ---------------------------------------------------------------------
    def checkpoint(self):
        """Everything that determines the handle's future: the state streams, the Philox seed and the tick."""
        ck = self.get_state()
        ck["seed"] = int(self.lib.soccer_get_seed(self.h)); ck["tick"] = self.tick
        ck["rng_abi"] = int(self.lib.soccer_abi_version())      # the bits -> uniform convention the (seed, tick) pair is meant for
        return ck

    def restore(self, ck):
        """Raises AssertionError for a checkpoint taken under another RNG convention (include/soccer_hip.h, SOCCER_ABI_VERSION:
        ABI 3 changed the bits -> uniform mapping, so the same (seed, tick) would continue on a different random stream); a
        checkpoint without the field predates ABI 3's field and is refused for the same reason."""
        abi = int(self.lib.soccer_abi_version())
        assert ck.get("rng_abi") == abi, \
            "checkpoint was taken under RNG ABI %r, this library is ABI %d: resuming would silently change the random stream" % (ck.get("rng_abi"), abi)
        self.set_state(ck["row_a"], ck["col_a"], ck["row_b"], ck["col_b"], ck["poss"], t=ck["t"],
                       needs_reset=ck["needs_reset"])
        self.seed(ck["seed"])
        self._check(self.lib.soccer_set_tick(self.h, int(ck["tick"])))

    # -- tables ---------------------------------------------------------------------------------
    def tables(self):
        lut = np.zeros(self.lut_len, np.uint16)
        goal_value = np.zeros(self.lut_len, np.int8)
        isd = np.zeros((self.n_isd, 5), np.int8)
        self._check(self.lib.soccer_get_tables(self.h, lut.ctypes.data, goal_value.ctypes.data, isd.ctypes.data))
        return lut, goal_value, isd

    def transitions(self):
        """The full transition relation (the reference's P_readable), enumerated on the device.
        Returns count[T,25] (-1: unreachable tuple), prob[T,25,36], next_flat[T,25,36], reward, done."""
        T = self.lut_len
        count = np.zeros((T, 25), np.int32); prob = np.zeros((T, 25, 36), np.float64)
        nxt = np.zeros((T, 25, 36), np.int32); rew = np.zeros((T, 25, 36), np.int8); done = np.zeros((T, 25, 36), np.uint8)
        self._check(self.lib.soccer_enumerate_transitions(self.h, count.ctypes.data, prob.ctypes.data,
                                                          nxt.ctypes.data, rew.ctypes.data, done.ctypes.data))
        return count, prob, nxt, rew, done

    # -- one environment, lowest latency (n_lanes == 1) ------------------------------------------------
    def step_scalar(self, state, act_a, act_b, u_step, t=0, u_reset=0.0):
        """soccer_step_scalar: (row_a, col_a, row_b, col_b, poss), actions, uniform -> dict with the next
        tuple, t, needs_reset, obs, reward, terminated, truncated, prob_code."""
        io = _lib.ScalarIO()
        io.row_a, io.col_a, io.row_b, io.col_b, io.poss = (int(x) for x in state)
        io.t = int(t); io.act_a = int(act_a or 0); io.act_b = int(act_b or 0)
        io.u_step = float(u_step); io.u_reset = float(u_reset)
        self._check(self.lib.soccer_step_scalar(self.h, C.byref(io)))
        return self._scalar_out(io)

    def reset_scalar(self, u_reset):
        io = _lib.ScalarIO()
        io.u_reset = float(u_reset)
        self._check(self.lib.soccer_reset_scalar(self.h, C.byref(io)))
        return self._scalar_out(io)

    @staticmethod
    def _scalar_out(io):
        return {"state": (io.row_a, io.col_a, io.row_b, io.col_b, io.poss), "t": io.t, "needs_reset": io.needs_reset,
                "obs": io.obs, "reward": io.reward, "terminated": io.terminated, "truncated": io.truncated,
                "prob_code": io.prob_code}

    # -- planners on the device (single-agent mode; reference utils/planners.py) ------------------
    def _plan_out(self):
        return (np.zeros(self.nS, np.float64), np.zeros((self.nS, 5), np.float64), np.zeros(self.nS, np.int32), C.c_int32())

    def value_iteration(self, theta, discount_factor, max_sweeps=1000000):
        """Best response of the learner against the handle's fixed policy.  Returns (pi, V, Q, iterations)
        like the reference's planners.value_iteration, bit for bit."""
        V, Q, pi, it = self._plan_out()
        self._check(self.lib.soccer_value_iteration(self.h, float(theta), float(discount_factor), int(max_sweeps),
                                                    V.ctypes.data, Q.ctypes.data, pi.ctypes.data, C.byref(it)))
        return pi.astype(np.int64), V, Q, int(it.value)

    def policy_evaluation(self, pi, theta, discount_factor, max_sweeps=1000000):
        pi = np.ascontiguousarray(np.asarray(pi).reshape(-1), np.int32)
        assert pi.shape == (self.nS,), "pi must have one action per observation index"
        V, _, _, it = self._plan_out()
        self._check(self.lib.soccer_policy_evaluation(self.h, pi.ctypes.data, float(theta), float(discount_factor),
                                                      int(max_sweeps), V.ctypes.data, C.byref(it)))
        return V, int(it.value)

    def policy_improvement(self, V, discount_factor):
        V = np.ascontiguousarray(np.asarray(V).reshape(-1), np.float64)
        assert V.shape == (self.nS,), "V must have one value per observation index"
        _, Q, pi, _ = self._plan_out()
        self._check(self.lib.soccer_policy_improvement(self.h, V.ctypes.data, float(discount_factor), Q.ctypes.data, pi.ctypes.data))
        return pi.astype(np.int64), Q

    def policy_iteration(self, pi0, theta, discount_factor, max_sweeps=1000000):
        pi0 = np.ascontiguousarray(np.asarray(pi0).reshape(-1), np.int32)
        assert pi0.shape == (self.nS,), "the initial policy must have one action per observation index"
        V, Q, pi, it = self._plan_out()
        self._check(self.lib.soccer_policy_iteration(self.h, pi0.ctypes.data, float(theta), float(discount_factor),
                                                     int(max_sweeps), V.ctypes.data, Q.ctypes.data, pi.ctypes.data, C.byref(it)))
        return pi.astype(np.int64), V, Q, int(it.value)

    def policy_eval_dense(self, policy, theta, discount_factor, k=10000000, init=None, max_sweeps=10000000):
        policy = np.ascontiguousarray(policy, np.float64)
        assert policy.shape == (self.nS, 5), "policy must be [nS, nA]"
        init_p = None
        if init is not None:
            init = np.ascontiguousarray(np.asarray(init).reshape(-1), np.float64)
            assert init.shape == (self.nS,)
            init_p = init.ctypes.data
        V, _, _, it = self._plan_out()
        self._check(self.lib.soccer_policy_eval_dense(self.h, policy.ctypes.data, int(min(k, 2**31 - 1)), float(theta),
                                                      float(discount_factor), int(max_sweeps), init_p, V.ctypes.data, C.byref(it)))
        return V, int(it.value)

    def modified_policy_iteration(self, k, theta, discount_factor, max_sweeps=10000000):
        V, Q, pi, it = self._plan_out()
        self._check(self.lib.soccer_modified_policy_iteration(self.h, int(min(k, 2**31 - 1)), float(theta), float(discount_factor),
                                                              int(max_sweeps), V.ctypes.data, Q.ctypes.data, pi.ctypes.data, C.byref(it)))
        return pi.astype(np.int64), V, Q, int(it.value)

    # -- hot path -------------------------------------------------------------------------------
    def reset(self, mask=None, u_reset=None, obs=None):
        self._check(self.lib.batched_reset(self.h, _ptr(mask), _ptr(u_reset), _ptr(obs)))

    def step(self, act_a, act_b, obs=None, reward=None, terminated=None, truncated=None,
             prob_code=None, u_step=None, u_reset=None, final_obs=None, last_return=None,
             reward_a_f32=None, reward_b_f32=None, finished=None):
        a = StepArgs(_ptr(act_a), _ptr(act_b), _ptr(u_step), _ptr(u_reset), _ptr(obs), _ptr(reward),
                     _ptr(terminated), _ptr(truncated), _ptr(prob_code), _ptr(final_obs), _ptr(last_return),
                     _ptr(reward_a_f32), _ptr(reward_b_f32), _ptr(finished))
        self._check(self.lib.batched_step_ex(self.h, C.byref(a)))

    def step_plain(self, act_a, act_b, obs, reward, terminated, truncated, prob_code=None):
        """The 8-argument batched_step entry point (per-lane Philox)."""
        self._check(self.lib.batched_step(self.h, _ptr(act_a), _ptr(act_b), _ptr(obs), _ptr(reward),
                                          _ptr(terminated), _ptr(truncated), _ptr(prob_code)))

    def rollout(self, n_steps, act_a=None, act_b=None, act_stride=0, sample_actions=False, obs=None,
                reward=None, terminated=None, truncated=None, out_stride=0, return_sum=None,
                episode_count=None, mix_a=None, mix_b=None, final_obs=None, prob_code=None):
        """batched_rollout; with final_obs / prob_code ([T][n] like the four result trajectories) batched_rollout_ex."""
        a = RolloutArgs(int(n_steps), 1 if sample_actions else 0, _ptr(act_a), _ptr(act_b), int(act_stride),
                        _ptr(obs), _ptr(reward), _ptr(terminated), _ptr(truncated), int(out_stride),
                        _ptr(return_sum), _ptr(episode_count), _ptr(mix_a), _ptr(mix_b))
        if final_obs is None and prob_code is None:
            self._check(self.lib.batched_rollout(self.h, C.byref(a)))
        else:
            x = _lib.RolloutExtra(_ptr(final_obs), _ptr(prob_code))
            self._check(self.lib.batched_rollout_ex(self.h, C.byref(a), C.byref(x)))

    def trajectory_returns(self, n_steps, reward, terminated, truncated, stride, last_return=None, episode_count=None, hist=True):
        """soccer_trajectory_returns: one pass over [n_steps][n] result trajectories (device) -> per-lane return of the most recently
        finished episode (int8[n]), per-lane finished-episode count (int32[n]) and, with hist=True (synchronises), the
        (-1, 0, +1) histogram of every finished episode as a uint64[3] array."""
        h3 = (C.c_uint64 * 3)() if hist else None
        self._check(self.lib.soccer_trajectory_returns(self.h, int(n_steps), _ptr(reward), _ptr(terminated), _ptr(truncated), int(stride),
                                                       _ptr(last_return), _ptr(episode_count), C.byref(h3) if hist else None))
        return np.array(h3, dtype=np.uint64) if hist else None

    # -- multi-GPU: RCCL over xGMI through the C ABI (one handle per process and GPU; see comm.py for the bring-up) -----
    def comm_init(self, world, rank, unique_id):
        assert len(unique_id) == _lib.COMM_ID_BYTES
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.soccer_comm_init(self.h, int(world), int(rank), buf))

    def comm_destroy(self):
        self._check(self.lib.soccer_comm_destroy(self.h))

    def all_gather(self, send, recv, bytes_per_rank):
        """recv[r * bytes_per_rank ...] = rank r's send[:bytes_per_rank] (device buffers; asynchronous on the handle's stream)."""
        self._check(self.lib.soccer_comm_all_gather(self.h, _ptr(send), _ptr(recv), int(bytes_per_rank)))

    def comm_sum(self, values):
        a = np.ascontiguousarray(values, np.uint64).copy()
        self._check(self.lib.soccer_comm_sum_u64(self.h, a.ctypes.data, int(a.size)))
        return a

    def comm_max(self, values):
        a = np.ascontiguousarray(values, np.float64).copy()
        self._check(self.lib.soccer_comm_max_f64(self.h, a.ctypes.data, int(a.size)))
        return a

    def comm_barrier(self):
        self._check(self.lib.soccer_comm_barrier(self.h))

    @staticmethod
    def mixed_policy_thresholds(probs):
        """[nS, 5] action probabilities -> uint16[nS, 4] cumulative thresholds for batched_rollout's
        mix_a / mix_b: floor(32768 * cumulative probability), values 0..32768; the sampled action is the
        number of thresholds <= a 15-bit draw, so deterministic rows are reproduced exactly."""
        p = np.asarray(probs, dtype=np.float64)
        assert p.ndim == 2 and p.shape[1] == 5 and (p >= 0).all() and np.allclose(p.sum(1), 1.0), \
            "probs must be [n_states, 5] rows summing to 1"
        c = np.cumsum(p, axis=1)[:, :4]
        return np.ascontiguousarray(np.clip(np.floor(c * 32768.0 + 1e-9), 0, 32768).astype(np.uint16))

    def set_policy(self, player, policy):
        """Fixed policy for 'player_a' / 'player_b' (dict or sequence: observation index -> action), or None."""
        idx = {"player_a": 0, "player_b": 1, 0: 0, 1: 1}[player]
        if policy is None:
            self._check(self.lib.soccer_set_policy(self.h, idx, None, 0)); return
        arr = np.array([policy[s] for s in range(self.nS)], dtype=np.int8) if isinstance(policy, dict) \
            else np.ascontiguousarray(policy, dtype=np.int8)
        self._check(self.lib.soccer_set_policy(self.h, idx, arr.ctypes.data, int(arr.size)))

    # -- zero-copy staged I/O: numpy views over the handle's pinned staging block ----------------
    def staging(self):
        """dict of numpy arrays (length n) over the pinned staging block: inputs act_a, act_b, mask, u_step,
        u_reset; outputs obs, final_obs, reward, terminated, truncated, prob_code (overwritten by the
        next staged call)."""
        if getattr(self, "_staging", None) is None:
            v = _lib.StagingView()
            self._check(self.lib.soccer_staging(self.h, C.byref(v)))
            def view(ptr, dt):
                dt = np.dtype(dt)
                buf = (C.c_uint8 * (self.n * dt.itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dt)
            self._staging = {"act_a": view(v.act_a, np.int8), "act_b": view(v.act_b, np.int8),
                             "mask": view(v.mask, np.uint8), "u_step": view(v.u_step, np.float64),
                             "u_reset": view(v.u_reset, np.float64), "obs": view(v.obs, np.uint16),
                             "final_obs": view(v.final_obs, np.uint16), "reward": view(v.reward, np.int8),
                             "terminated": view(v.terminated, np.uint8), "truncated": view(v.truncated, np.uint8),
                             "prob_code": view(v.prob_code, np.uint8)}
        return self._staging

    def step_staged(self, act_a=True, act_b=True, u_step=False, u_reset=False):
        use = (_lib.STAGE_ACT_A if act_a else 0) | (_lib.STAGE_ACT_B if act_b else 0) | \
              (_lib.STAGE_U_STEP if u_step else 0) | (_lib.STAGE_U_RESET if u_reset else 0)
        self._check(self.lib.batched_step_staged(self.h, use))

    def reset_staged(self, mask=False, u_reset=False):
        self._check(self.lib.batched_reset_staged(self.h, (_lib.STAGE_MASK if mask else 0) |
                                                  (_lib.STAGE_U_RESET if u_reset else 0)))

    # -- host-array variants (numpy in, numpy out; one staged copy each way) -------------------
    def reset_host(self, mask=None, u_reset=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        u = None if u_reset is None else np.ascontiguousarray(u_reset, np.float64)
        obs = np.empty(self.n, np.uint16)
        self._check(self.lib.batched_reset_host(self.h, None if m is None else m.ctypes.data,
                                                None if u is None else u.ctypes.data, obs.ctypes.data))
        return obs

    def step_host(self, act_a, act_b, u_step=None, u_reset=None):
        n = self.n
        a = None if act_a is None else np.ascontiguousarray(act_a, np.int8)
        b = None if act_b is None else np.ascontiguousarray(act_b, np.int8)
        assert (a is None or a.shape == (n,)) and (b is None or b.shape == (n,)), \
            "actions must have one entry per environment"
        us = None if u_step is None else np.ascontiguousarray(u_step, np.float64)
        ur = None if u_reset is None else np.ascontiguousarray(u_reset, np.float64)
        out = {"obs": np.empty(n, np.uint16), "reward": np.empty(n, np.int8),
               "terminated": np.empty(n, np.uint8), "truncated": np.empty(n, np.uint8),
               "prob_code": np.empty(n, np.uint8), "final_obs": np.empty(n, np.uint16)}
        args = StepArgs(None if a is None else a.ctypes.data, None if b is None else b.ctypes.data, None if us is None else us.ctypes.data,
                        None if ur is None else ur.ctypes.data, out["obs"].ctypes.data,
                        out["reward"].ctypes.data, out["terminated"].ctypes.data,
                        out["truncated"].ctypes.data, out["prob_code"].ctypes.data,
                        out["final_obs"].ctypes.data, None)
        self._check(self.lib.batched_step_host(self.h, C.byref(args)))
        return out

    def host_state_view(self):
        """host_mapped handles: numpy uint8 view [6, n] over the pinned state streams (row_a, col_a, row_b,
        col_b, poss | needs_reset << 1, t) the GPU works on in place.  Touch it only while the stream is idle."""
        p = C.c_void_p(); stride = C.c_uint64()
        self._check(self.lib.soccer_host_view(self.h, C.byref(p), C.byref(stride)))
        buf = (C.c_uint8 * (6 * stride.value)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(6, stride.value)[:, :self.n]

    # -- state injection / readback -----------------------------------------------------------
    def set_state(self, row_a=None, col_a=None, row_b=None, col_b=None, poss=None, t=None, needs_reset=None):
        def arr(x, dt):
            if x is None:
                return None, None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dt), (self.n,)), dt)
            return a, a.ctypes.data
        keep = []
        ptrs = []
        for x, dt in ((row_a, np.int8), (col_a, np.int8), (row_b, np.int8), (col_b, np.int8),
                      (poss, np.uint8), (t, np.uint8), (needs_reset, np.uint8)):
            a, p = arr(x, dt); keep.append(a); ptrs.append(p)
        code = self.lib.soccer_set_state(self.h, *ptrs)
        if code == _lib.E_INVALID:
            msg = self.lib.soccer_last_error(self.h).decode()
            if "not a reachable state tuple" in msg:
                raise KeyError(msg)        # the reference's P_readable[self.state] lookup (:394)
        self._check(code)

    def get_state(self):
        n = self.n
        out = {k: np.zeros(n, np.int8) for k in ("row_a", "col_a", "row_b", "col_b")}
        out.update({k: np.zeros(n, np.uint8) for k in ("poss", "t", "needs_reset")})
        self._check(self.lib.soccer_get_state(self.h, *[out[k].ctypes.data for k in
                    ("row_a", "col_a", "row_b", "col_b", "poss", "t", "needs_reset")]))
        return out

    # -- statistics / timing ------------------------------------------------------------------
    def stats(self):
        hist = (C.c_uint64 * 3)(); mis = C.c_uint64()
        self._check(self.lib.soccer_get_stats(self.h, C.byref(hist), C.byref(mis)))
        return np.array(hist, dtype=np.uint64), int(mis.value)

    MISUSE_FROZEN, MISUSE_ACTION = 1, 2

    def misuse(self):
        """The sticky misuse flags alone (no histogram copy; synchronises): MISUSE_FROZEN if a lane was stepped while
        it needed reset (:376), MISUSE_ACTION if a device-side action byte was outside 0..4 (:393)."""
        mis = C.c_uint64()
        self._check(self.lib.soccer_get_stats(self.h, None, C.byref(mis)))
        return int(mis.value)

    def peek_misuse(self):
        """The same flags WITHOUT synchronising: what the launches completed so far have raised."""
        return int(self.lib.soccer_peek_misuse(self.h))

    def reset_stats(self):
        self._check(self.lib.soccer_reset_stats(self.h))

    def timer_start(self):
        self._check(self.lib.soccer_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.soccer_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def timer_mark(self):
        """Record the closing event (allowed inside a graph capture, like timer_start)."""
        self._check(self.lib.soccer_timer_mark(self.h))

    def timer_read(self):
        ms = C.c_float()
        self._check(self.lib.soccer_timer_read(self.h, C.byref(ms)))
        return float(ms.value)

    def stamp(self, slot):
        """Enqueue (or capture) a device clock stamp into slot 0..255 of the handle's host-mapped block."""
        self._check(self.lib.soccer_stamp(self.h, int(slot)))

    def stamps_clear(self, first=0, count=256):
        self._check(self.lib.soccer_stamps_clear(self.h, int(first), int(count)))

    def stamps(self, first=0, count=256):
        """(ticks[count] uint64, clock kHz) — the slots as they stand, no synchronisation; 0 = not written since cleared."""
        t = np.zeros(int(count), np.uint64); khz = C.c_int32()
        self._check(self.lib.soccer_stamps_read(self.h, int(first), int(count), t.ctypes.data, C.byref(khz)))
        return t, int(khz.value)

    # -- hipGraph capture ----------------------------------------------------------------------
    def graph_begin(self):
        self._check(self.lib.soccer_graph_begin(self.h))

    def graph_end(self):
        g = C.c_void_p()
        self._check(self.lib.soccer_graph_end(self.h, C.byref(g)))
        return g

    def graph_launch(self, g, replays=1):
        self._check(self.lib.soccer_graph_launch(self.h, g, int(replays)))

    def graph_destroy(self, g):
        self.lib.soccer_graph_destroy(self.h, g)
