"""ctypes driver for the CPU oracle (``oracle/libfw_oracle.so``).

TEST INFRASTRUCTURE ONLY -- see the header of ``fw_oracle.c``.  Importable from
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg;
never from the product package.  It takes the config as an opaque ctypes
structure (anything with the memory layout of ``include/fwsim.h:fw_config``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfw_oracle.so")
FW_STATE_DIM = 176
FW_INFO_DIM = 8


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fw_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fwsim.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfw_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_fast() -> str:
    """-O3 -march=native build of the SAME C for bench.py's cpu_baseline leg.  -march=native binds the binary to the
    CPU it was compiled on, so it is rebuilt whenever the host CPU model differs from the one recorded beside it."""
    path = os.path.join(_HERE, "libfw_oracle_fast.so")
    tag = path + ".cpu"
    src = os.path.join(_HERE, "fw_oracle.c")
    model = _cpu_model()
    ok = os.path.exists(path) and os.path.exists(tag) and open(tag).read() == model and os.path.getmtime(path) >= os.path.getmtime(src)
    if not ok:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfw_oracle_fast.so"], stdout=subprocess.DEVNULL)
        with open(tag, "w") as f:
            f.write(model)
    return path


_lib = None
_fast = None


def fast_lib() -> C.CDLL:
    global _fast
    if _fast is None:
        _fast = _bind(C.CDLL(build_fast()))
    return _fast


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L: C.CDLL) -> C.CDLL:
    vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
    L.fwo_sizeof_config.restype = i32
    L.fwo_abi_version.restype = i32
    L.fwo_state_dim.restype = i32
    L.fwo_obs_dim.restype = i32; L.fwo_obs_dim.argtypes = [vp]
    L.fwo_validate_config.restype = i32; L.fwo_validate_config.argtypes = [vp, C.c_char_p, i32]
    L.fwo_create.restype = i32; L.fwo_create.argtypes = [vp, i32, i32, u64, i64, C.POINTER(vp)]
    L.fwo_reset.restype = i32; L.fwo_reset.argtypes = [vp, vp, vp, vp, vp]
    L.fwo_step.restype = i32; L.fwo_step.argtypes = [vp] * 9
    L.fwo_observe.restype = i32; L.fwo_observe.argtypes = [vp, vp, vp]
    L.fwo_seed.restype = i32; L.fwo_seed.argtypes = [vp, u64]
    L.fwo_get_state.restype = i32; L.fwo_get_state.argtypes = [vp, vp]
    L.fwo_set_state.restype = i32; L.fwo_set_state.argtypes = [vp, vp]
    L.fwo_set_threads.restype = i32; L.fwo_set_threads.argtypes = [i32]
    L.fwo_render.restype = i32; L.fwo_render.argtypes = [vp, i32, vp, vp]
    L.fwo_num_envs.restype = i32; L.fwo_num_envs.argtypes = [vp]
    L.fwo_last_error.restype = C.c_char_p; L.fwo_last_error.argtypes = [vp]
    L.fwo_destroy.restype = i32; L.fwo_destroy.argtypes = [vp]
    L.fwo_aero_coeffs.restype = None; L.fwo_aero_coeffs.argtypes = [vp, C.c_double, C.c_double, vp]
    L.fwo_surface_constants.restype = None; L.fwo_surface_constants.argtypes = [vp, vp]
    L.fwo_surface_force.restype = None; L.fwo_surface_force.argtypes = [vp, i32, C.c_double, vp, vp, vp]
    L.fwo_euler_from_quat.restype = None; L.fwo_euler_from_quat.argtypes = [vp, vp]
    L.fwo_quat_from_euler.restype = None; L.fwo_quat_from_euler.argtypes = [vp, vp]
    L.fwo_mat_from_quat.restype = None; L.fwo_mat_from_quat.argtypes = [vp, vp]
    L.fwo_philox.restype = None; L.fwo_philox.argtypes = [vp, vp, vp]
    L.fwo_rng_uniform01.restype = C.c_double; L.fwo_rng_uniform01.argtypes = [u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.fwo_rng_normal2.restype = None; L.fwo_rng_normal2.argtypes = [u64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.fwo_wind_at.restype = None; L.fwo_wind_at.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp]
    L.fwo_depth_buffer_to_meters.restype = C.c_double; L.fwo_depth_buffer_to_meters.argtypes = [C.c_double]
    if L.fwo_state_dim() != FW_STATE_DIM:
        raise RuntimeError("oracle FW_STATE_DIM does not match the Python wrapper")
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """N scalar CPU envs behind the same call shapes as the HIP library."""

    def __init__(self, cfg, num_envs: int, seed: int = 0, global_env_offset: int = 0, fast: bool = False):
        self._cfg = cfg
        L = self._L = fast_lib() if fast else lib()
        if L.fwo_sizeof_config() != C.sizeof(cfg):
            raise RuntimeError("fw_config layout mismatch between Python mirror and oracle")
        h = C.c_void_p()
        rc = L.fwo_create(C.byref(cfg), int(num_envs), 0, int(seed), int(global_env_offset), C.byref(h))
        if rc != 0:
            msg = L.fwo_last_error(None).decode()
            raise ValueError(msg) if rc in (-1, -4) else RuntimeError(msg)
        self._h = h
        self.num_envs = int(num_envs)
        self.obs_dim = int(L.fwo_obs_dim(C.byref(cfg)))
        self.dtype = np.float64 if cfg.dtype == 0 else np.float32

    def close(self):
        if self._h:
            self._L.fwo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None, scenario=None) -> np.ndarray:
        """`scenario`: an fw_scenario-shaped ctypes structure (host arrays; the caller keeps them alive) or None."""
        obs = np.empty((self.num_envs, self.obs_dim), dtype=self.dtype)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = self._L.fwo_reset(self._h, _ptr(m), None if scenario is None else C.byref(scenario), _ptr(obs), None)
        assert rc == 0, self._L.fwo_last_error(self._h)
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=self.dtype).reshape(self.num_envs, 4)
        n, d = self.num_envs, self.obs_dim
        obs = np.empty((n, d), dtype=self.dtype)
        rew = np.empty((n,), dtype=self.dtype)
        term = np.empty((n,), dtype=np.uint8)
        trunc = np.empty((n,), dtype=np.uint8)
        tobs = np.zeros((n, d), dtype=self.dtype)
        info = np.empty((n, FW_INFO_DIM), dtype=np.int32)
        rc = self._L.fwo_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc), _ptr(tobs), _ptr(info), None)
        assert rc == 0
        return obs, rew, term, trunc, tobs, info

    def step_timed_only(self, actions, obs, rew, term, trunc, info):
        """Step into caller-provided buffers (bench leg; no allocation)."""
        return self._L.fwo_step(self._h, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc), None, _ptr(info), None)

    def observe(self) -> np.ndarray:
        obs = np.empty((self.num_envs, self.obs_dim), dtype=self.dtype)
        assert self._L.fwo_observe(self._h, _ptr(obs), None) == 0
        return obs

    def render(self, res: int) -> np.ndarray:
        """fw_render twin: float32 [N, 2, res, res] = (duck mask, depth buffer) of every env's current pose, pixel by pixel."""
        img = np.empty((self.num_envs, 2, int(res), int(res)), dtype=np.float32)
        rc = self._L.fwo_render(self._h, int(res), _ptr(img), None)
        assert rc == 0, self._L.fwo_last_error(self._h)
        return img

    def seed(self, seed: int):
        assert self._L.fwo_seed(self._h, int(seed)) == 0

    def get_state(self) -> np.ndarray:
        s = np.empty((self.num_envs, FW_STATE_DIM), dtype=np.float64)
        assert self._L.fwo_get_state(self._h, _ptr(s)) == 0
        return s

    def set_state(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64).reshape(self.num_envs, FW_STATE_DIM)
        assert self._L.fwo_set_state(self._h, _ptr(s)) == 0


def set_threads(n: int, fast: bool = False) -> int:
    """Set (n>0) / query (n<=0) the OpenMP thread count used by fwo_step."""
    return int((fast_lib() if fast else lib()).fwo_set_threads(int(n)))


# ---- unit-level helpers for known-answer tests ----
def aero_coeffs(surface_params, alpha: float, defl: float) -> np.ndarray:
    out = np.empty(3)
    lib().fwo_aero_coeffs(C.byref(surface_params), float(alpha), float(defl), _ptr(out))
    return out


def surface_constants(surface_params) -> np.ndarray:
    out = np.empty(5)
    lib().fwo_surface_constants(C.byref(surface_params), _ptr(out))
    return out


def surface_force(cfg, s: int, actuation: float, v_local):
    v = np.ascontiguousarray(v_local, dtype=np.float64)
    f, t = np.empty(3), np.empty(3)
    lib().fwo_surface_force(C.byref(cfg), int(s), float(actuation), _ptr(v), _ptr(f), _ptr(t))
    return f, t


def euler_from_quat(q) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64); e = np.empty(3)
    lib().fwo_euler_from_quat(_ptr(q), _ptr(e)); return e


def quat_from_euler(e) -> np.ndarray:
    e = np.ascontiguousarray(e, dtype=np.float64); q = np.empty(4)
    lib().fwo_quat_from_euler(_ptr(e), _ptr(q)); return q


def mat_from_quat(q) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64); m = np.empty(9)
    lib().fwo_mat_from_quat(_ptr(q), _ptr(m)); return m.reshape(3, 3)


def philox(ctr, key) -> np.ndarray:
    c = np.ascontiguousarray(ctr, dtype=np.uint32); k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib().fwo_philox(_ptr(c), _ptr(k), _ptr(o)); return o


def rng_uniform01(seed, env, ep, stream, j) -> float:
    return float(lib().fwo_rng_uniform01(int(seed), int(env), int(ep), int(stream), int(j)))


def rng_normal2(seed, env, ep, astep) -> np.ndarray:
    z = np.empty(2)
    lib().fwo_rng_normal2(int(seed), int(env), int(ep), int(astep), _ptr(z)); return z


def wind_at(cfg, base, amp, phase: float, t: float) -> np.ndarray:
    b = np.ascontiguousarray(base, dtype=np.float64); a = np.ascontiguousarray(amp, dtype=np.float64)
    w = np.empty(3)
    lib().fwo_wind_at(C.byref(cfg), _ptr(b), _ptr(a), float(phase), float(t), _ptr(w)); return w


def depth_buffer_to_meters(d: float) -> float:
    return float(lib().fwo_depth_buffer_to_meters(float(d)))
