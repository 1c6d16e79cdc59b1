"""Action / observation space objects of the env mirrors.

The reference builds `gymnasium.spaces.Box` / `Discrete` objects (rayleigh/rayleigh.py:75-86, mixing/mixing.py:61-70,
burgers/burgers.py:54-65, shkadov/shkadov.py:96-110, sloshing/sloshing.py:73-86, lorenz/lorenz.py:49-57,
vortex/vortex.py:69-79).  When gymnasium is importable the mirrors carry the real thing -- a trainer's
`isinstance(env.action_space, gymnasium.spaces.Box)` then holds -- with the reference's bounds, shapes and dtype
(float32); gymnasium is optional (it is not installable in the build container), so without it the stand-ins below
provide the attributes the reference's users read (`shape`, `low`, `high`, `n`, `dtype`, `sample()`)."""
import numpy as np

try:
    from gymnasium import spaces as _gsp
except ImportError:                      # optional dependency
    _gsp = None


class Box(object):
    """Stand-in for gymnasium.spaces.Box (used only when gymnasium is not importable)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        shape = tuple(np.shape(low)) if shape is None else tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        self.shape, self.dtype = shape, np.dtype(dtype)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(object):
    """Stand-in for gymnasium.spaces.Discrete."""

    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return int(rng.integers(0, self.n))

    def contains(self, x):
        return int(x) == x and 0 <= int(x) < self.n


def have_gymnasium():
    return _gsp is not None


def box(low, high, shape):
    """Box(low, high, shape, float32) exactly as the reference constructs it: `low` / `high` scalars (action spaces) or
    arrays of `shape` (observation spaces)."""
    cls = _gsp.Box if _gsp is not None else Box
    return cls(low=low, high=high, shape=tuple(shape), dtype=np.float32)


def sym_box(bound, n):
    """The reference's observation spaces: high = bound * ones(n), Box(-high, high, (n,), float32)."""
    high = np.ones(int(n)) * bound
    return box(-high, high, (int(n),))


def discrete(n):
    return (_gsp.Discrete if _gsp is not None else Discrete)(int(n))
