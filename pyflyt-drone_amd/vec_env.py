"""SB3-``VecEnv``-shaped host mirror of the reference's env stack for the hot path.

In the reference the stack is ``SubprocVecEnv([make_env(i) ...])`` over
``FlattenWaypointEnv(gym.make("PyFlyt/Fixedwing-Waypoints-v3", ...))``
(train/train_Fixedwing_Waypoints_v3.py:82-121,251).  Here the N envs are one
HIP kernel launch; this class only owns device tensors and forwards to the
C ABI (include/fwsim.h).  Two surfaces:

* numpy / SB3 ``VecEnv`` duck type: ``reset() -> obs[N,D]``,
  ``step_async(actions)``, ``step_wait() -> (obs, rewards, dones, infos)``,
  ``step``, ``close``, ``seed``, ``get_attr/set_attr/env_method/env_is_wrapped``
  with the worker semantics SB3 implements (auto-reset, ``terminal_observation``,
  ``TimeLimit.truncated``; info keys of
  envs/fixedwing_envs/fixedwing_base_env.py:212-215 and ``num_targets_reached``).
* device fast path: ``reset_tensor()`` / ``step_tensor(actions)`` return torch
  tensors living on the GPU; nothing touches the host.
"""
from __future__ import annotations

import ctypes as C
from copy import deepcopy
from typing import Any, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import config as K
from .spaces import Box


def _devptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


try:        # Stable-Baselines3 asserts isinstance(env, VecEnv): subclass the real base class when it is importable
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv as _VecEnvBase
    HAVE_SB3 = True
except Exception:
    HAVE_SB3 = False

    class _VecEnvBase:
        """In-tree mirror of ``stable_baselines3.common.vec_env.VecEnv`` (constructor contract and the concrete helpers
        SB3's algorithms call) for images without SB3 -- the build image is one."""

        def __init__(self, num_envs: int, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space, self.action_space = observation_space, action_space
            self.reset_infos = [{} for _ in range(num_envs)]
            self._seeds = [None for _ in range(num_envs)]
            self._options = [{} for _ in range(num_envs)]
            modes = self.get_attr("render_mode")
            assert all(m == modes[0] for m in modes), "render_mode mode should be the same for all environments"
            self.render_mode = modes[0]
            self.metadata = {"render_modes": [] if self.render_mode is None else [self.render_mode]}

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()

        def seed(self, seed=None):
            if seed is None:
                seed = int(np.random.randint(0, np.iinfo(np.uint32).max, dtype=np.uint32))
            self._seeds = [seed + idx for idx in range(self.num_envs)]
            return self._seeds

        def set_options(self, options=None):
            options = {} if options is None else options
            self._options = deepcopy(options) if isinstance(options, list) else deepcopy([options] * self.num_envs)

        def _reset_seeds(self):
            self._seeds = [None for _ in range(self.num_envs)]

        def _reset_options(self):
            self._options = [{} for _ in range(self.num_envs)]

        @property
        def unwrapped(self):
            return self

        def getattr_depth_check(self, name: str, already_found: bool):
            return None

        def _get_indices(self, indices):
            if indices is None:
                return range(self.num_envs)
            return [indices] if isinstance(indices, int) else indices


class FixedwingVecEnv(_VecEnvBase):
    """N fixed-wing envs advanced in lockstep by ``fw_step`` on one MI355X.  A ``stable_baselines3`` ``VecEnv``
    (subclass of the real base class when SB3 is importable, of its in-tree mirror otherwise) whose spaces are
    ``gymnasium.spaces.Box`` when gymnasium is importable."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 30}

    def __init__(self, cfg: K.FwConfig, num_envs: int, device: Optional[int | str | torch.device] = None,
                 seed: int = 0, global_env_offset: int = 0):
        _lib.validate(cfg)
        if num_envs <= 0:
            raise ValueError("num_envs must be positive")
        if not torch.cuda.is_available():
            raise RuntimeError("pyflyt_drone_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.cfg = cfg.copy()
        self.num_envs = int(num_envs)
        self.global_env_offset = int(global_env_offset)
        self.seed_value = int(seed)
        self.obs_dim = K.obs_dim(cfg)
        self.np_dtype = np.float64 if cfg.dtype == K.FW_F64 else np.float32
        self.torch_dtype = torch.float64 if cfg.dtype == K.FW_F64 else torch.float32
        self.render_mode = None
        _VecEnvBase.__init__(self, self.num_envs, Box(-np.inf, np.inf, (self.obs_dim,), self.np_dtype),
                             Box(-1.0, 1.0, (4,), self.np_dtype))

        h = C.c_void_p()
        rc = _lib.lib().fw_create(C.byref(self.cfg), self.num_envs, int(dev.index), int(seed) & (2**64 - 1),
                                  int(global_env_offset), C.byref(h))
        _lib.check(rc, None)
        self._h = h
        self.lanes_per_env = int(_lib.lib().fw_lanes_per_env(h))    # which lane mapping fw_create picked (diagnostic)
        self.g8_waves = 2 if self.lanes_per_env == 16 else 1
        self.capture_wave = bool(_lib.lib().fw_capture_wave(h))      # camera tasks: step workgroups with a capture wave (FWSIM_CAPTURE_WAVE=1)
        if self.lanes_per_env == 16:                                # (16 = the 8-lane mapping built for two waves per SIMD)
            self.lanes_per_env = 8
        n, d = self.num_envs, self.obs_dim
        kw = dict(device=dev)
        self.obs = torch.zeros((n, d), dtype=self.torch_dtype, **kw)
        self.rewards = torch.zeros((n,), dtype=self.torch_dtype, **kw)
        self.terminated = torch.zeros((n,), dtype=torch.uint8, **kw)
        self.truncated = torch.zeros((n,), dtype=torch.uint8, **kw)
        self.terminal_obs = torch.zeros((n, d), dtype=self.torch_dtype, **kw)
        self.info = torch.zeros((n, K.FW_INFO_DIM), dtype=torch.int32, **kw)
        self._actions_dev = torch.zeros((n, 4), dtype=self.torch_dtype, **kw)
        self._pending = False

    # ------------------------------------------------------------------ device fast path
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset_tensor(self, mask: Optional[torch.Tensor] = None, scenario: Optional[dict] = None) -> torch.Tensor:
        """Reset all envs (or those with ``mask != 0``) on the device; returns obs[N,D].  ``scenario``: keyword
        arguments of :func:`config.make_scenario` (targets / duck_pos / obstacles + num_obstacles / wind_base / gust_amp /
        gust_phase as host arrays) that replace what the env would draw for the episodes started here."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.num_envs:
                raise ValueError("mask must have num_envs elements")
        sc, keep = (None, None) if scenario is None else K.make_scenario(self.num_envs, **scenario)
        rc = _lib.lib().fw_reset(self._h, _devptr(mask), None if sc is None else C.byref(sc), _devptr(self.obs), self._stream())
        del keep
        _lib.check(rc, self._h)
        return self.obs

    def step_tensor(self, actions: torch.Tensor):
        """One agent step.  ``actions``: device tensor [N,4] of the env dtype in [-1,1]
        (the caller clips, as SB3's collector does).  Returns views of the env-owned
        output tensors ``(obs, rewards, terminated, truncated)``; ``terminal_obs`` and
        ``info`` are attributes.  No host synchronisation."""
        if actions.device != self.device or actions.dtype != self.torch_dtype or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=self.torch_dtype).contiguous()
        if actions.shape != (self.num_envs, 4):
            raise ValueError(f"actions must have shape ({self.num_envs}, 4), got {tuple(actions.shape)}")
        rc = _lib.lib().fw_step(self._h, _devptr(actions), _devptr(self.obs), _devptr(self.rewards),
                                _devptr(self.terminated), _devptr(self.truncated), _devptr(self.terminal_obs),
                                _devptr(self.info), self._stream())
        _lib.check(rc, self._h)
        return self.obs, self.rewards, self.terminated, self.truncated

    def observe_tensor(self) -> torch.Tensor:
        rc = _lib.lib().fw_observe(self._h, _devptr(self.obs), self._stream())
        _lib.check(rc, self._h)
        return self.obs

    def render_tensor(self, res: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """FPV image of every env's current pose: float32 ``[N, 2, res, res]`` = (duck mask, depth buffer) of the analytic
        scene the vision features are functionals of (``fw_render``; what ``Camera.capture_image()`` hands the reference env,
        envs/fixedwing_objlock_env.py:603-622).  Camera tasks only."""
        res = int(res)
        if out is None:
            out = torch.empty((self.num_envs, 2, res, res), dtype=torch.float32, device=self.device)
        if out.shape != (self.num_envs, 2, res, res) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("render_tensor: out must be a contiguous float32 [N, 2, res, res] tensor on the env's device")
        rc = _lib.lib().fw_render(self._h, res, _devptr(out), self._stream())
        _lib.check(rc, self._h)
        return out

    # ------------------------------------------------------------------ SB3 VecEnv surface (numpy)
    def reset(self) -> np.ndarray:
        obs = self.reset_tensor()
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return self._np_obs(obs)

    def _np_obs(self, t: torch.Tensor) -> np.ndarray:
        """Host copy of an observation tensor in the dtype `observation_space` declares (ObjLock: the reference's flattened
        observation is float32, envs/flatten_objlock_env.py:29-31,46; the values are already float32-rounded on the device)."""
        a = t.cpu().numpy()
        want = self.observation_space.dtype
        return a if a.dtype == want else a.astype(want)

    def step_async(self, actions: np.ndarray) -> None:
        a = torch.as_tensor(np.asarray(actions), dtype=self.torch_dtype).reshape(self.num_envs, 4)
        self._actions_dev.copy_(a, non_blocking=False)
        self.step_tensor(self._actions_dev)
        self._pending = True

    def step_wait(self):
        if not self._pending:
            raise RuntimeError("step_wait() called without step_async()")
        self._pending = False
        obs = self._np_obs(self.obs)
        rewards = self.rewards.cpu().numpy()
        term = self.terminated.cpu().numpy().astype(bool)
        trunc = self.truncated.cpu().numpy().astype(bool)
        info = self.info.cpu().numpy()
        dones = term | trunc
        infos: List[dict] = []
        tobs = self._np_obs(self.terminal_obs) if dones.any() else None
        for i in range(self.num_envs):
            d = {
                "out_of_bounds": bool(info[i, K.INFO_OUT_OF_BOUNDS]),
                "collision": bool(info[i, K.INFO_COLLISION]),
                "env_complete": bool(info[i, K.INFO_ENV_COMPLETE]),
                "num_targets_reached": int(info[i, K.INFO_NUM_TARGETS_REACHED]),
                "TimeLimit.truncated": bool(trunc[i] and not term[i]),
            }
            if self.cfg.task != K.FW_TASK_WAYPOINTS:
                d["duck_strike"] = bool(info[i, K.INFO_DUCK_STRIKE])
                d["is_success"] = bool(info[i, K.INFO_IS_SUCCESS])
            if dones[i] and self.cfg.auto_reset:
                d["terminal_observation"] = tobs[i].copy()
                d["episode_length"] = int(info[i, K.INFO_EP_LEN])
            infos.append(d)
        return obs, rewards, dones, infos

    def step(self, actions: np.ndarray):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed: Optional[int] = None) -> Sequence[Optional[int]]:
        s = 0 if seed is None else int(seed)
        _lib.check(_lib.lib().fw_seed(self._h, s & (2**64 - 1)), self._h)
        self.seed_value = s
        self._seeds = [s + i for i in range(self.num_envs)]
        return self._seeds

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().fw_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _indices(self, indices) -> List[int]:
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    # Per-env attribute access of the VecEnv API.  The N envs are one kernel launch with one config, so every "per-env"
    # attribute is the same for all of them: the env object's own attributes first, then the fields of its fw_config
    # (flight_dome_size, num_targets, ...: what the reference env objects carry as attributes).
    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        if attr_name == "render_mode":
            v = self.__dict__.get("render_mode", None)
        elif attr_name in self.__dict__ or hasattr(type(self), attr_name):
            v = getattr(self, attr_name)
        elif "cfg" in self.__dict__ and any(attr_name == f[0] for f in type(self.cfg)._fields_):
            v = getattr(self.cfg, attr_name)
        else:
            raise AttributeError(f"{type(self).__name__} envs have no attribute {attr_name!r}")
        return [v for _ in self._indices(indices)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        """Host-side attributes can be set (for all envs at once: they share one object); anything that is compiled into
        the device-side constants (an ``fw_config`` field) needs a new env."""
        if "cfg" in self.__dict__ and any(attr_name == f[0] for f in type(self.cfg)._fields_):
            raise AttributeError(f"{attr_name!r} is part of the device-side configuration; build a new env with it")
        if indices is not None and sorted(self._indices(indices)) != list(range(self.num_envs)):
            raise AttributeError("the envs of a fused device env share their attributes: set them for all envs (indices=None)")
        setattr(self, attr_name, value)

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        """Call a method "of each env": methods of this object are called ONCE (they already act on all envs) and the
        result is repeated per requested index, as ``SubprocVecEnv.env_method`` would return it."""
        fn = getattr(self, method_name, None)
        if not callable(fn) or method_name.startswith("_"):
            raise AttributeError(f"env_method({method_name!r}) is not available on a fused device env")
        out = fn(*args, **kwargs)
        return [out for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode: Optional[str] = None):
        return None

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ state access (parity tests / checkpoints)
    def get_state(self) -> np.ndarray:
        s = np.empty((self.num_envs, K.FW_STATE_DIM), dtype=np.float64)
        _lib.check(_lib.lib().fw_get_state(self._h, s.ctypes.data_as(C.c_void_p)), self._h)
        return s

    def get_counters(self) -> dict:
        """How the auto-resets of ``fw_step`` were served so far (``fw_get_counters``): launches, resets and whether
        they took the background hand-off (shadow / pre-sampled scenario) or the in-kernel fallback."""
        c = np.zeros(K.FW_CTR_DIM, dtype=np.uint64)
        _lib.check(_lib.lib().fw_get_counters(self._h, c.ctypes.data_as(C.c_void_p)), self._h)
        return {"launches": int(c[K.CTR_LAUNCHES]), "resets": int(c[K.CTR_RESETS]), "shadow_hits": int(c[K.CTR_SHADOW_HITS]),
                "scenario_hits": int(c[K.CTR_SCENARIO_HITS]), "fallbacks": int(c[K.CTR_FALLBACKS]),
                "capture_wave_timeouts": int(c[K.CTR_HELPER_TIMEOUTS])}

    def set_state(self, state: np.ndarray) -> None:
        s = np.ascontiguousarray(state, dtype=np.float64).reshape(self.num_envs, K.FW_STATE_DIM)
        _lib.check(_lib.lib().fw_set_state(self._h, s.ctypes.data_as(C.c_void_p)), self._h)


class FixedwingWaypointsVecEnv(FixedwingVecEnv):
    """``PyFlyt/Fixedwing-Waypoints-v3`` + ``FlattenWaypointEnv``, vectorised.

    Keyword arguments are those of the upstream env constructor as passed at
    train/train_Fixedwing_Waypoints_v3.py:100-110 plus the wrapper's
    ``context_length`` (:117) and the wind dict of ``WindOnResetWrapper`` (:112-113).
    """

    def __init__(self, num_envs: int, *, sparse_reward: bool = False, num_targets: int = 4,
                 goal_reach_distance: float = 2.0, flight_dome_size: float = 100.0,
                 max_duration_seconds: float = 120.0, angle_representation: str = "quaternion",
                 agent_hz: int = 30, context_length: int = 2, wind_config: Optional[dict] = None,
                 render_mode: Optional[str] = None, dtype: str = "float64", motor_noise: bool = True,
                 device=None, seed: int = 0, global_env_offset: int = 0):
        if render_mode is not None:
            raise ValueError(f"Invalid render mode {render_mode}, rendering is not part of the device env.")
        cfg = K.waypoints_config(sparse_reward=sparse_reward, num_targets=num_targets,
                                 goal_reach_distance=goal_reach_distance, flight_dome_size=flight_dome_size,
                                 max_duration_seconds=max_duration_seconds,
                                 angle_representation=angle_representation, agent_hz=agent_hz,
                                 context_length=context_length, wind_config=wind_config, dtype=dtype,
                                 motor_noise=motor_noise)
        super().__init__(cfg, num_envs, device=device, seed=seed, global_env_offset=global_env_offset)


class FixedwingObjLockVecEnv(FixedwingVecEnv):
    """``FlattenObjLockEnv(FixedwingObjLockEnv(...))`` vectorised (envs/fixedwing_objlock_env.py:37-81,
    envs/flatten_objlock_env.py; constructed at train/train_objlock.py:113-153).  Accepts the reference
    constructor's full keyword set (``config.objlock_config_from_reference_kwargs``: render-only arguments are ignored,
    options the device env cannot honour raise ``ValueError``).  Observations are the reference's 56 float32 values (52 with ``duck_vision_use_deltas=False``)
    (22 attitude + 3 target vector + 31 duck vision); with ``dtype="float64"`` they are stored in float64 tensors
    but already rounded to float32."""

    def __init__(self, num_envs: int, *, dtype: str = "float64", motor_noise: bool = True, device=None, seed: int = 0,
                 global_env_offset: int = 0, **env_kwargs):
        cfg = K.objlock_config_from_reference_kwargs(dtype=dtype, motor_noise=motor_noise, **env_kwargs)
        super().__init__(cfg, num_envs, device=device, seed=seed, global_env_offset=global_env_offset)
        self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,), np.float32)


class FixedwingWaypointObjLockVecEnv(FixedwingVecEnv):
    """``FlattenWaypointEnv(FixedwingWaypointObjLockEnv(...), context_length)`` vectorised
    (envs/fixedwing_waypoint_objlock_env.py:42-76; constructed at
    train/train_Fixedwing_Waypoints_ObjLock.py:119-165; full reference keyword set, see
    ``config.waypoint_objlock_config_from_reference_kwargs``).  Observation = attitude ++ the first
    ``context_length`` rows of [remaining waypoints ..., duck] in the body frame (float64)."""

    def __init__(self, num_envs: int, *, context_length: int = 2, dtype: str = "float64", motor_noise: bool = True,
                 device=None, seed: int = 0, global_env_offset: int = 0, **env_kwargs):
        cfg = K.waypoint_objlock_config_from_reference_kwargs(dtype=dtype, motor_noise=motor_noise,
                                                              context_length=context_length, **env_kwargs)
        super().__init__(cfg, num_envs, device=device, seed=seed, global_env_offset=global_env_offset)
