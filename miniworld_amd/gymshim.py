"""Gymnasium surface used by the package.

If the real ``gymnasium`` is importable it is used unchanged.  Otherwise (this image has no
gymnasium and no network) a minimal API-compatible stand-in is provided so that
``reset(seed=...)`` / ``step`` / ``spaces`` / ``register`` / ``make`` keep working:
``Env.reset(seed)`` seeds ``np_random`` exactly like gymnasium.utils.seeding.np_random
(``Generator(PCG64(SeedSequence(seed)))``), which the reference relies on (miniworld.py:551).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as gym
    from gymnasium import spaces
    from gymnasium.utils import EzPickle
    HAVE_GYMNASIUM = True
    VectorEnvBase = gym.vector.VectorEnv                # miniworld_amd.vector.MiniWorldVectorEnv is one when gymnasium is there
    try:                                                # gymnasium >= 1.1: an enum; before: the plain string
        from gymnasium.vector import AutoresetMode
        AUTORESET_SAME_STEP = AutoresetMode.SAME_STEP
    except ImportError:
        AUTORESET_SAME_STEP = "same-step"

    def batch_action_space(single, n):
        from gymnasium.vector.utils import batch_space
        return batch_space(single, n)
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False
    VectorEnvBase = object
    AUTORESET_SAME_STEP = "same-step"

    def batch_action_space(single, n):
        return Box(single.start, single.start + single.n - 1, (n,), dtype=np.int64)

    class _Space:
        def __init__(self, shape=None, dtype=None):
            self.shape, self.dtype = shape, (np.dtype(dtype) if dtype is not None else None)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__((), np.int64)
            self.n, self.start = int(n), int(start)

        def sample(self):
            return int(self._rng.integers(self.start, self.start + self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__(tuple(shape) if shape is not None else np.shape(low), dtype)
            self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape)
            self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape)

        def sample(self):
            if np.issubdtype(self.dtype, np.integer):
                return self._rng.integers(self.low, self.high, endpoint=True, dtype=self.dtype)
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(_Space, dict):
        def __init__(self, spaces_=None, **kw):
            _Space.__init__(self)
            dict.__init__(self, spaces_ or {}, **kw)

    class _SpacesModule:
        Discrete, Box, Dict = Discrete, Box, Dict

    spaces = _SpacesModule()

    class EzPickle:
        def __init__(self, *args, **kwargs):
            self._ezpickle_args, self._ezpickle_kwargs = args, kwargs

        def __getstate__(self):
            return {"_ezpickle_args": self._ezpickle_args, "_ezpickle_kwargs": self._ezpickle_kwargs}

        def __setstate__(self, d):
            out = type(self)(*d["_ezpickle_args"], **d["_ezpickle_kwargs"])
            self.__dict__.update(out.__dict__)

    class Env:
        metadata: dict = {}
        render_mode = None
        _np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
            return self._np_random

        @np_random.setter
        def np_random(self, value):
            self._np_random = value

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class _Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

        @property
        def np_random(self):            # gymnasium.Wrapper forwards the rng to the wrapped env
            return self.env.np_random

        @np_random.setter
        def np_random(self, value):
            self.env.np_random = value

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def reset(self, **kw):
            return self.env.reset(**kw)

        def step(self, action):
            return self.env.step(action)

    class ObservationWrapper(_Wrapper):
        def reset(self, **kw):
            obs, info = self.env.reset(**kw)
            return self.observation(obs), info

        def step(self, action):
            obs, r, te, tr, info = self.env.step(action)
            return self.observation(obs), r, te, tr, info

    class ActionWrapper(_Wrapper):
        def step(self, action):
            return self.env.step(self.action(action))

    class _Logger:
        @staticmethod
        def warn(msg, *a):
            import warnings
            warnings.warn(msg % a if a else msg)

    class _GymModule:
        Env, spaces, Wrapper = Env, spaces, _Wrapper
        ObservationWrapper, ActionWrapper = ObservationWrapper, ActionWrapper
        logger = _Logger()
        _registry: dict = {}

        @classmethod
        def register(cls, id, entry_point, **kwargs):  # noqa: A002
            cls._registry[id] = (entry_point, kwargs)

        @classmethod
        def make(cls, id, **kwargs):  # noqa: A002
            import importlib
            if id not in cls._registry:
                raise KeyError(f"unknown environment id {id!r}")
            entry, base_kwargs = cls._registry[id]
            if isinstance(entry, str):
                mod, name = entry.split(":")
                entry = getattr(importlib.import_module(mod), name)
            kw = dict(base_kwargs.get("kwargs", {}))
            kw.update(kwargs)
            return entry(**kw)

    gym = _GymModule()
