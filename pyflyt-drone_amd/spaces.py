"""``Box`` space of the env surface.

When ``gymnasium`` is importable, ``Box`` IS ``gymnasium.spaces.Box`` -- Stable-Baselines3 asserts
``isinstance(observation_space, gymnasium.spaces.Box)`` on what it is handed, so the reference's
``PPO("MlpPolicy", env, ...)`` (train/train_Fixedwing_Waypoints_v3.py:293) accepts these envs as they are.  gymnasium is
not a dependency of this package (and is absent from the build image): otherwise ``Box`` is the in-tree mirror below with
what SB3-style callers read from ``observation_space`` / ``action_space`` (``shape``, ``dtype``, ``low``, ``high``,
``sample()``, ``contains()``).  Mirrors the spaces declared at envs/fixedwing_envs/fixedwing_base_env.py:74-94 and
envs/flatten_waypoint_env.py:45-50.
"""
from __future__ import annotations

import numpy as np

try:                                        # the real class when the caller's environment has it
    from gymnasium.spaces import Box as _GymBox
except Exception:                           # ModuleNotFoundError here; any import-time failure means "not usable"
    _GymBox = None


class MirrorBox:
    def __init__(self, low, high, shape=None, dtype=np.float64, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self) -> np.ndarray:
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __contains__(self, x) -> bool:
        return self.contains(x)

    def __repr__(self) -> str:
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other) -> bool:
        return (isinstance(other, type(self)) and self.shape == other.shape and self.dtype == other.dtype
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))


Box = _GymBox if _GymBox is not None else MirrorBox
HAVE_GYMNASIUM = _GymBox is not None
