"""ctypes binding of libsoccer_hip.so (include/soccer_hip.h).

There is no Python or CPU fallback for the hot path: if the HIP library is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsoccer_hip.so")

OK, E_INVALID, E_HIP, E_NOMEM, E_STATE = 0, -1, -2, -3, -4
F_AUTORESET = 1
F_NULL_STREAM = 2
F_HOST_MAPPED = 4
F_STEP_STATS = 8
F_STREAM_ACTIONS = 16


class SoccerHipError(RuntimeError):
    """A HIP runtime failure inside libsoccer_hip.so."""


class Config(C.Structure):
    _fields_ = [("n_lanes", C.c_uint64), ("width", C.c_int32), ("height", C.c_int32),
                ("slip_prob", C.c_double), ("max_steps", C.c_int32), ("device", C.c_int32),
                ("seed", C.c_uint64), ("lane_offset", C.c_uint64), ("flags", C.c_uint32),
                ("envs_per_thread", C.c_uint32), ("stream", C.c_void_p)]


class StepArgs(C.Structure):
    _fields_ = [("act_a", C.c_void_p), ("act_b", C.c_void_p), ("u_step", C.c_void_p),
                ("u_reset", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p),
                ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("prob_code", C.c_void_p),
                ("final_obs", C.c_void_p), ("last_return", C.c_void_p),
                ("reward_a_f32", C.c_void_p), ("reward_b_f32", C.c_void_p), ("finished", C.c_void_p)]


class RolloutArgs(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("sample_actions", C.c_int32), ("act_a", C.c_void_p),
                ("act_b", C.c_void_p), ("act_stride", C.c_int64), ("obs", C.c_void_p),
                ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
                ("out_stride", C.c_int64), ("return_sum", C.c_void_p), ("episode_count", C.c_void_p),
                ("mix_a", C.c_void_p), ("mix_b", C.c_void_p)]


class RolloutExtra(C.Structure):
    """soccer_rollout_extra"""
    _fields_ = [("final_obs", C.c_void_p), ("prob_code", C.c_void_p)]


class ScalarIO(C.Structure):
    """soccer_scalar_io"""
    _fields_ = [("row_a", C.c_int8), ("col_a", C.c_int8), ("row_b", C.c_int8), ("col_b", C.c_int8),
                ("poss", C.c_uint8), ("needs_reset", C.c_uint8), ("t", C.c_uint8),
                ("act_a", C.c_int8), ("act_b", C.c_int8), ("reward", C.c_int8),
                ("terminated", C.c_uint8), ("truncated", C.c_uint8), ("prob_code", C.c_uint8), ("pad_", C.c_uint8),
                ("obs", C.c_uint16), ("u_step", C.c_double), ("u_reset", C.c_double)]


class StagingView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("act_a", "act_b", "mask", "u_step", "u_reset", "obs", "final_obs",
                                         "reward", "terminated", "truncated", "prob_code")]


STAGE_ACT_A, STAGE_ACT_B, STAGE_U_STEP, STAGE_U_RESET, STAGE_MASK = 1, 2, 4, 8, 16
COMM_ID_BYTES = 128

# name -> (restype, argtypes); every symbol include/soccer_hip.h declares
PROTOTYPES = {
    "soccer_abi_version": (C.c_int, []),
    "soccer_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "soccer_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "soccer_destroy": (C.c_int, [C.c_void_p]),
    "soccer_last_error": (C.c_char_p, [C.c_void_p]),
    "soccer_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "soccer_sync": (C.c_int, [C.c_void_p]),
    "batched_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "batched_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "batched_step_ex": (C.c_int, [C.c_void_p, C.POINTER(StepArgs)]),
    "batched_rollout": (C.c_int, [C.c_void_p, C.POINTER(RolloutArgs)]),
    "batched_rollout_ex": (C.c_int, [C.c_void_p, C.POINTER(RolloutArgs), C.POINTER(RolloutExtra)]),
    "batched_step_host": (C.c_int, [C.c_void_p, C.POINTER(StepArgs)]),
    "batched_reset_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "soccer_staging": (C.c_int, [C.c_void_p, C.POINTER(StagingView)]),
    "batched_step_staged": (C.c_int, [C.c_void_p, C.c_uint32]),
    "batched_reset_staged": (C.c_int, [C.c_void_p, C.c_uint32]),
    "soccer_set_policy": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "soccer_host_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "soccer_set_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "soccer_get_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "soccer_dims": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 4),
    "soccer_get_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "soccer_enumerate_transitions": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5),
    "soccer_step_scalar": (C.c_int, [C.c_void_p, C.POINTER(ScalarIO)]),
    "soccer_reset_scalar": (C.c_int, [C.c_void_p, C.POINTER(ScalarIO)]),
    "soccer_value_iteration": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.POINTER(C.c_int32)]),
    "soccer_policy_evaluation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                           C.POINTER(C.c_int32)]),
    "soccer_policy_improvement": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "soccer_policy_iteration": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "soccer_policy_eval_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "soccer_modified_policy_iteration": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "soccer_prob_table": (C.c_int, [C.c_void_p, C.POINTER(C.c_double * 12)]),
    "soccer_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_uint64)]),
    "soccer_reset_stats": (C.c_int, [C.c_void_p]),
    "soccer_peek_misuse": (C.c_uint32, [C.c_void_p]),
    "soccer_tick": (C.c_uint64, [C.c_void_p]),
    "soccer_get_seed": (C.c_uint64, [C.c_void_p]),
    "soccer_set_tick": (C.c_int, [C.c_void_p, C.c_uint64]),
    "soccer_trajectory_returns": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64 * 3)]),
    "soccer_comm_unique_id": (C.c_int, [C.c_void_p]),
    "soccer_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "soccer_comm_destroy": (C.c_int, [C.c_void_p]),
    "soccer_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "soccer_comm_sum_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "soccer_comm_max_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "soccer_comm_barrier": (C.c_int, [C.c_void_p]),
    "soccer_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "soccer_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "soccer_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "soccer_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "soccer_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "soccer_timer_start": (C.c_int, [C.c_void_p]),
    "soccer_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "soccer_timer_mark": (C.c_int, [C.c_void_p]),
    "soccer_timer_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "soccer_stamp": (C.c_int, [C.c_void_p, C.c_int32]),
    "soccer_stamps_clear": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "soccer_stamps_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]),
    "soccer_graph_begin": (C.c_int, [C.c_void_p]),
    "soccer_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "soccer_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "soccer_graph_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and HSA runtime.
    Two HIP runtimes in one process cannot both open the GPU, so when torch is installed make sure
    the copy torch will use is the one already in the process before libsoccer_hip.so asks for
    libamdhip64.so.7: the loader then resolves both to the same runtime, and device pointers,
    streams and events are interchangeable between torch and this library.
    Set SOCCER_HIP_RUNTIME=system to skip (standalone use without torch)."""
    import importlib.util
    import sys
    if os.environ.get("SOCCER_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libsoccer_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsoccer_hip.so not found at %s — build it with "
                "`make -C gym_soccer_littman94_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback for the batched step/reset path." % LIB_PATH)
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(lib, handle, code):
    """Turn a C-ABI error code into the exception the reference would raise."""
    if code == OK:
        return
    msg = lib.soccer_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else "unknown error"
    if code == E_INVALID:
        raise AssertionError(msg)           # the reference's error convention is plain assert
    if code == E_NOMEM:
        raise MemoryError(msg)
    if code == E_STATE:
        raise RuntimeError(msg)
    raise SoccerHipError(msg)
