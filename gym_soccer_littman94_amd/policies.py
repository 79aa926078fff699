"""Fixed policies for single-agent mode — the reference's `gym_soccer.utils.policies`
(gym_soccer/utils/policies.py:4-27): dicts observation index -> action."""
import numpy as np

NOOP = 0


def get_random_policy(n_states=761, n_actions=5, seed=0):               # policies.py:4-9
    random_state = np.random.RandomState(seed)
    return {s: random_state.randint(0, n_actions) for s in range(n_states)}


def get_stand_policy(n_states=761):                                      # policies.py:11-15
    return {s: NOOP for s in range(n_states)}


def save_policy(policy, filename, mode='wb'):                            # policies.py:17-22
    import pickle
    assert isinstance(policy, dict), "Policy must be a dictionary"
    with open(filename, mode) as f:
        pickle.dump({int(k): int(v) for k, v in policy.items()}, f)


def load_policy(filename, mode='rb'):                                    # policies.py:24-27
    """Reads a file written by save_policy (a pickle: only open files you wrote yourself)."""
    import pickle
    with open(filename, mode) as f:
        return pickle.load(f)
