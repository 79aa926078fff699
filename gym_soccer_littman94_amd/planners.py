"""Planners on the device — the reference's `gym_soccer.utils.planners` (gym_soccer/utils/planners.py:4-87)
with the same names, arguments and return values, each running as ONE HIP kernel over transition lists
enumerated on the device (libsoccer_hip.so, include/soccer_hip.h "planners").

Like the reference's they need a single-agent env (one side with a fixed policy).  The list-based planners
(`value_iteration`, `policy_evaluation`, `policy_improvement`, `policy_iteration`) are bit-identical to the
reference's float64 loops; the dense ones (`policy_eval`, `modified_policy_iteration`) follow its Pmat/Rmat
algebra and agree to rounding (numpy's BLAS dot sums in another order).  `env` is a
`SoccerSimultaneousEnv`, a `VectorSoccerEnv` or a `SoccerBatch` of this package.
"""
import numpy as np


def _batch(env):
    from .core import SoccerBatch
    if isinstance(env, SoccerBatch):
        return env
    b = getattr(env, "_batch", None)
    if b is None:
        raise TypeError("planners expect a gym_soccer_littman94_amd environment")
    assert not env.multiagent, "planners need a single-agent environment (one player with a fixed policy)"
    return b


def value_iteration(env, theta, discount_factor):                        # planners.py:4-18
    return _batch(env).value_iteration(theta, discount_factor)


def policy_evaluation(pi, env, theta, discount_factor):                  # planners.py:20-31
    return _batch(env).policy_evaluation(pi, theta, discount_factor)[0]


def policy_improvement(V, env, discount_factor):                         # planners.py:33-41
    return _batch(env).policy_improvement(V, discount_factor)


def policy_iteration(env, theta, discount_factor, initial_policy=None):  # planners.py:43-53
    b = _batch(env)
    if initial_policy is None:                 # the reference's draw, from numpy's global generator (:45)
        initial_policy = np.random.choice((0, 1, 2, 3, 4), b.nS)
    return b.policy_iteration(initial_policy, theta, discount_factor)


def policy_eval(env, policy, theta, discount_factor, k=10000000, init=None):   # planners.py:55-70
    v, cc = _batch(env).policy_eval_dense(policy, theta, discount_factor, k=k, init=init)
    if init is not None:
        init[:] = v                            # the reference updates `init` in place (v[:] = value_fc) and returns it
        v = init
    return v, cc


def modified_policy_iteration(env, k, theta, discount_factor):           # planners.py:73-87
    return _batch(env).modified_policy_iteration(k, theta, discount_factor)
