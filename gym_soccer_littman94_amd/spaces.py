"""observation_space / action_space objects.

The reference builds `gym.spaces.Dict({agent: gym.spaces.Discrete(n)})`
(gym_soccer/envs/soccer_simultaneous_env.py:126-131).  `gym` / `gymnasium` are optional here: when
one is importable its classes are used, otherwise these minimal stand-ins with the same `.n`,
`.sample()`, `.contains()` and mapping behaviour.
"""
import numpy as np


def _backend():
    for name in ("gymnasium", "gym"):
        try:
            mod = __import__(name)
            return mod.spaces
        except Exception:
            continue
    return None


class _Discrete:
    def __init__(self, n, seed=None):
        self.n = int(n)
        self.dtype = np.int64
        self.shape = ()
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        try:
            return 0 <= int(x) < self.n and int(x) == x
        except Exception:
            return False

    __contains__ = contains

    def __eq__(self, other):
        return hasattr(other, "n") and int(other.n) == self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class _MultiDiscrete:
    """N independent Discrete(n) — the batched space of the vector env."""
    def __init__(self, nvec, seed=None):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.int64
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.integers(0, self.nvec)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((x >= 0) & (x < self.nvec)).all())

    __contains__ = contains

    def __repr__(self):
        return "MultiDiscrete(%d x %d)" % (self.nvec.size, int(self.nvec.flat[0]) if self.nvec.size else 0)


class _Dict(dict):
    def __init__(self, spaces=None, **kw):
        super().__init__(spaces or {}, **kw)
        self.spaces = self

    def sample(self):
        return {k: s.sample() for k, s in self.items()}

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.keys() and all(self[k].contains(v) for k, v in x.items())

    def __repr__(self):
        return "Dict(%s)" % ", ".join("%s: %r" % kv for kv in self.items())


_b = _backend()
Discrete = _b.Discrete if _b is not None else _Discrete
Dict = _b.Dict if _b is not None else _Dict
MultiDiscrete = _b.MultiDiscrete if _b is not None else _MultiDiscrete
