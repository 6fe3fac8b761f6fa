"""ctypes binding of ``csrc/libfwsim_hip.so`` (the C ABI of ``include/fwsim.h``).

There is no CPU fallback: if the shared library is missing, or no HIP device
is present, env construction raises.  Building is done by
``__graft_entry__.build()`` / :func:`build` with ``hipcc --offload-arch=gfx950``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import config as K

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FWSIM_LIB") or os.path.join(CSRC, "libfwsim_hip.so")      # FWSIM_LIB: A/B builds of the same ABI (dev tools)
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

EXPORTS = (
    "fw_sizeof_config", "fw_abi_version", "fw_state_dim", "fw_obs_dim", "fw_validate_config", "fw_create", "fw_reset",
    "fw_step", "fw_seed", "fw_get_state", "fw_set_state", "fw_get_counters", "fw_observe", "fw_render", "fw_num_envs", "fw_lanes_per_env", "fw_capture_wave", "fw_last_error",
    "fw_destroy", "fw_gae", "fw_eval_track", "fw_normalize_obs", "fw_normalize_obs_workspace_bytes", "fw_ppo_update_workspace_bytes", "fw_ppo_param_count", "fw_ppo_moment_count", "fw_ppo_moment_map", "fw_ppo_update", "fw_policy_act", "fw_policy_terminal_value", "fw_rollout_post", "fw_collect_act", "fw_collect_stats", "fw_collect_stats_workspace_bytes", "fw_collect_step", "fw_collect_finish", "fw_collect_step_workspace_bytes", "fw_collect_workspace_init", "fw_collect_close", "fw_collect_status", "fw_ppo_update_status",
)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
    deps = srcs + [os.path.join(INCLUDE, "fwsim.h")]
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(p) for p in deps)
    if force or stale:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        # -save-temps (into a scratch directory): the device assembly of exactly this build is checked for a register-allocator
        # hazard that silently corrupts lanes (tools/check_isa.py) -- a build that has it must not ship
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp(prefix="fwsim_build_")
        try:
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-save-temps",
                   "-o", LIB_PATH, os.path.join(CSRC, "fwsim.hip")]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=tmp)
            asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
            if not asm:
                raise RuntimeError("hipcc left no gfx950 assembly to check")
            hits = check_isa(open(os.path.join(tmp, asm[0])).read())
            if hits:
                os.remove(LIB_PATH)
                raise RuntimeError("ISA check failed (tools/check_isa.py: spill stores before the exec restore of a join block, "
                                   "or a last-block-done ticket that can overtake its partial sums): "
                                   + "; ".join(f"{n[:60]} {lab}" for n, lab, _ in hits[:6]))
            if verbose:
                print("ISA check: no spill store precedes the exec restore of its block; ticket atomics sit behind a vmcnt(0) wait")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return LIB_PATH


def check_isa(text: str):
    """tools/check_isa.py, importable from the package's own build step."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fw_check_isa", os.path.join(os.path.dirname(CSRC.rstrip(os.sep)), "..", "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.scan(text) + mod.scan_ticket(text)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "pyflyt_drone_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
        L.fw_sizeof_config.restype = i32; L.fw_sizeof_config.argtypes = []
        L.fw_abi_version.restype = i32; L.fw_abi_version.argtypes = []
        L.fw_state_dim.restype = i32; L.fw_state_dim.argtypes = []
        L.fw_obs_dim.restype = i32; L.fw_obs_dim.argtypes = [vp]
        L.fw_validate_config.restype = i32; L.fw_validate_config.argtypes = [vp, C.c_char_p, i32]
        L.fw_create.restype = i32; L.fw_create.argtypes = [vp, i32, i32, u64, i64, C.POINTER(vp)]
        L.fw_reset.restype = i32; L.fw_reset.argtypes = [vp, vp, vp, vp, vp]
        L.fw_step.restype = i32; L.fw_step.argtypes = [vp] * 9
        L.fw_observe.restype = i32; L.fw_observe.argtypes = [vp, vp, vp]
        L.fw_seed.restype = i32; L.fw_seed.argtypes = [vp, u64]
        L.fw_get_state.restype = i32; L.fw_get_state.argtypes = [vp, vp]
        L.fw_set_state.restype = i32; L.fw_set_state.argtypes = [vp, vp]
        L.fw_get_counters.restype = i32; L.fw_get_counters.argtypes = [vp, vp]
        L.fw_num_envs.restype = i32; L.fw_num_envs.argtypes = [vp]
        L.fw_lanes_per_env.restype = i32; L.fw_lanes_per_env.argtypes = [vp]
        L.fw_capture_wave.restype = i32; L.fw_capture_wave.argtypes = [vp]
        L.fw_render.restype = i32; L.fw_render.argtypes = [vp, i32, vp, vp]
        L.fw_last_error.restype = C.c_char_p; L.fw_last_error.argtypes = [vp]
        L.fw_destroy.restype = i32; L.fw_destroy.argtypes = [vp]
        L.fw_gae.restype = i32
        L.fw_eval_track.restype = i32
        L.fw_eval_track.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
        L.fw_gae.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_float, vp]
        L.fw_normalize_obs.restype = i32
        L.fw_normalize_obs.argtypes = [vp, i32, i32, i32, vp, vp, vp, i32, C.c_float, C.c_float, vp, vp, vp, vp]
        L.fw_normalize_obs_workspace_bytes.restype = i64; L.fw_normalize_obs_workspace_bytes.argtypes = [i32]
        L.fw_ppo_update_workspace_bytes.restype = i64; L.fw_ppo_update_workspace_bytes.argtypes = [i32, i32, i32]
        L.fw_ppo_param_count.restype = i32; L.fw_ppo_param_count.argtypes = [i32]
        L.fw_ppo_moment_count.restype = i32; L.fw_ppo_moment_count.argtypes = []
        L.fw_ppo_moment_map.restype = i32; L.fw_ppo_moment_map.argtypes = [i32, vp]
        L.fw_ppo_update.restype = i32
        L.fw_ppo_update.argtypes = [vp] * 9 + [i32, i32, i32, vp, vp, vp, i64, vp]
        L.fw_policy_act.restype = i32
        L.fw_policy_act.argtypes = [vp, vp, i32, i32, i32, i32, vp, i64, vp, vp, vp, i32, vp, vp, vp]
        L.fw_policy_terminal_value.restype = i32
        L.fw_policy_terminal_value.argtypes = [vp, vp, i32, i32, i32, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp]
        f32, f64 = C.c_float, C.c_double
        L.fw_collect_act.restype = i32
        L.fw_collect_act.argtypes = [vp, vp, i32, i32, i32, vp, vp, f32, f32, i32, i32, vp, i64, vp, vp, vp, i32, vp, vp,
                                     vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp]
        L.fw_collect_stats_workspace_bytes.restype = i64; L.fw_collect_stats_workspace_bytes.argtypes = [i32]
        L.fw_collect_stats.restype = i32
        L.fw_collect_stats.argtypes = [vp, i32, i32, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, f64, vp, vp, vp, vp, vp]
        L.fw_collect_step_workspace_bytes.restype = i64; L.fw_collect_step_workspace_bytes.argtypes = [vp]
        L.fw_collect_step.restype = i32; L.fw_collect_step.argtypes = [vp, vp, vp]
        L.fw_collect_finish.restype = i32; L.fw_collect_finish.argtypes = [vp, vp, vp]
        L.fw_collect_workspace_init.restype = i32; L.fw_collect_workspace_init.argtypes = [vp, vp, i64, vp]
        L.fw_collect_close.restype = i32; L.fw_collect_close.argtypes = [vp, vp, vp, vp]
        L.fw_collect_status.restype = i32; L.fw_collect_status.argtypes = [vp, vp, i64, vp, vp]
        L.fw_ppo_update_status.restype = i32; L.fw_ppo_update_status.argtypes = [vp, i64, vp, vp, vp]
        L.fw_rollout_post.restype = i32
        L.fw_rollout_post.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_double, C.c_float, C.c_float, vp, vp, vp, vp, vp]
        if L.fw_abi_version() != K.FW_ABI_VERSION:
            raise RuntimeError("libfwsim_hip.so ABI version does not match the Python binding; rebuild")
        if L.fw_state_dim() != K.FW_STATE_DIM:
            raise RuntimeError("FW_STATE_DIM mismatch between include/fwsim.h and config.py")
        if L.fw_sizeof_config() != C.sizeof(K.FwConfig):
            raise RuntimeError("fw_config layout mismatch between include/fwsim.h and config.FwConfig")
        _lib = L
    return _lib


def check(rc: int, handle=None) -> None:
    """Map ABI error codes onto the reference's exception types
    (``ValueError`` for bad configuration, ``RuntimeError`` otherwise)."""
    if rc == K.FW_OK:
        return
    msg = lib().fw_last_error(handle)
    msg = msg.decode() if msg else f"fwsim error {rc}"
    if rc in (K.FW_EINVAL, K.FW_EVERSION):
        raise ValueError(msg)
    if rc == K.FW_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def validate(cfg: K.FwConfig) -> None:
    buf = C.create_string_buffer(256)
    rc = lib().fw_validate_config(C.byref(cfg), buf, 256)
    if rc != K.FW_OK:
        raise ValueError(buf.value.decode())
