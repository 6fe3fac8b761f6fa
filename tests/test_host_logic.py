"""CPU-only checks of the host side: the C-ABI library loads and exports every
symbol include/fwsim.h declares, config validation mirrors the reference's
ValueErrors, and the product fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pyflyt_drone_amd as P
from pyflyt_drone_amd import _lib
from pyflyt_drone_amd import config as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fwsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fw_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/fwsim.h but not exported"


def test_config_layout_matches_header():
    L = _lib.lib()
    assert L.fw_sizeof_config() == C.sizeof(K.FwConfig)
    assert L.fw_abi_version() == K.FW_ABI_VERSION
    cfg = K.train_waypoints_v3_config()
    assert L.fw_obs_dim(C.byref(cfg)) == 28 == K.obs_dim(cfg)
    cfg = K.waypoints_config(angle_representation="quaternion", context_length=3)
    assert L.fw_obs_dim(C.byref(cfg)) == 13 + 4 + 6 + 9


def test_header_enums_match_python_mirror():
    text = open(os.path.join(ROOT, "include", "fwsim.h")).read()
    vals = dict((k, int(v)) for k, v in re.findall(r"\b(FW_[A-Z_0-9]+)\s*=\s*(-?\d+)", text))
    vals.update((k, int(v)) for k, v in re.findall(r"#define\s+(FW_[A-Z_0-9]+)\s+(\d+)", text))
    assert vals["FW_ABI_VERSION"] == K.FW_ABI_VERSION and vals["FW_STATE_DIM"] == K.FW_STATE_DIM
    for py, h in [("S_POS", "FW_S_POS"), ("S_QUAT", "FW_S_QUAT"), ("S_VEL", "FW_S_VEL"), ("S_OMEGA", "FW_S_OMEGA"),
                  ("S_ACT", "FW_S_ACT"), ("S_ACTION", "FW_S_ACTION"), ("S_STEP_COUNT", "FW_S_STEP_COUNT"),
                  ("S_TICK_COUNT", "FW_S_TICK_COUNT"), ("S_EPISODE", "FW_S_EPISODE"), ("S_FLAGS", "FW_S_FLAGS"),
                  ("S_NUM_REACHED", "FW_S_NUM_REACHED"), ("S_NEW_DIST", "FW_S_NEW_DIST"), ("S_WIND", "FW_S_WIND"),
                  ("S_EP_RETURN", "FW_S_EP_RETURN"), ("S_TARGETS", "FW_S_TARGETS"), ("S_TASK", "FW_S_TASK"),
                  ("INFO_NUM_TARGETS_REACHED", "FW_INFO_NUM_TARGETS_REACHED"), ("INFO_COLLISION", "FW_INFO_COLLISION"),
                  ("INFO_OUT_OF_BOUNDS", "FW_INFO_OUT_OF_BOUNDS"), ("INFO_ENV_COMPLETE", "FW_INFO_ENV_COMPLETE"),
                  ("INFO_EP_LEN", "FW_INFO_EP_LEN"), ("FW_INFO_DIM", "FW_INFO_DIM"),
                  ("FW_MAX_TARGETS", "FW_MAX_TARGETS"), ("FW_EHIP", "FW_EHIP"), ("FW_EINVAL", "FW_EINVAL")]:
        assert getattr(K, py) == vals[h], (py, h)


def test_validate_config_through_the_abi():
    cfg = K.train_waypoints_v3_config()
    _lib.validate(cfg)
    bad = cfg.copy(); bad.agent_hz = 50
    with pytest.raises(ValueError, match="try 40 or 60"):
        _lib.validate(bad)
    bad = cfg.copy(); bad.angle_representation = 3
    with pytest.raises(ValueError, match="euler"):
        _lib.validate(bad)
    bad = cfg.copy(); bad.wind_mode = 7
    with pytest.raises(ValueError, match="Unsupported wind mode"):
        _lib.validate(bad)
    bad = cfg.copy(); bad.abi_version = 1
    with pytest.raises(ValueError, match="abi_version"):
        _lib.validate(bad)
    bad = cfg.copy(); bad.num_targets = 99
    with pytest.raises(ValueError, match="num_targets"):
        _lib.validate(bad)


def test_validation_messages_agree_with_oracle(oracle):
    for mutate in (lambda c: setattr(c, "agent_hz", 7), lambda c: setattr(c, "angle_representation", 2),
                   lambda c: setattr(c, "wind_mode", -1), lambda c: setattr(c, "mass", 0.0)):
        c = K.train_waypoints_v3_config(); mutate(c)
        buf_a, buf_b = C.create_string_buffer(256), C.create_string_buffer(256)
        ra = _lib.lib().fw_validate_config(C.byref(c), buf_a, 256)
        rb = oracle.lib().fwo_validate_config(C.byref(c), buf_b, 256)
        assert ra == rb == K.FW_EINVAL and buf_a.value == buf_b.value


def test_python_constructor_errors_mirror_reference():
    with pytest.raises(ValueError, match="`agent_hz` must be round denominator of 120"):
        K.waypoints_config(agent_hz=50)
    with pytest.raises(ValueError, match="angle_representation must be either `euler` or `quaternion`"):
        K.waypoints_config(angle_representation="rpy")
    with pytest.raises(ValueError, match="num_targets"):
        K.waypoints_config(num_targets=9)
    with pytest.raises(ValueError, match="dtype"):
        K.waypoints_config(dtype="bf16")


def _same_config(a, b):
    import ctypes as C
    return C.string_at(C.addressof(a), C.sizeof(a)) == C.string_at(C.addressof(b), C.sizeof(b))


def test_reference_make_env_keyword_sets_are_accepted():
    """The exact keyword sets of the reference's make_env() factories (train/train_objlock.py:113-153 and
    train/train_Fixedwing_Waypoints_ObjLock.py:119-165) build the training configs; what the device env cannot honour
    raises ValueError instead of TypeError / silent acceptance."""
    objlock_kw = dict(
        sparse_reward=False, render_mode="rgb_array", angle_representation="euler", flight_dome_size=200.0,
        max_duration_seconds=60.0, agent_hz=30, use_egl=False, wind_config=K.TRAIN_OBJLOCK_WIND,
        num_obstacles=0, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0), obstacle_safe_distance_m=10.0,
        obstacle_avoid_reward_scale=1.0, obstacle_avoid_max_penalty=5.0,
        duck_camera_capture_interval_steps=12, duck_lock_hold_steps=5, duck_strike_distance_m=10.0, duck_strike_reward=400.0,
        duck_lock_step_reward=0.2, duck_approach_reward_scale=0.1, duck_global_scaling=60.0,
        duck_vision_history_len=3, duck_vision_use_deltas=True)
    c = K.objlock_config_from_reference_kwargs(**objlock_kw)
    assert _same_config(c, K.train_objlock_config()) and c.camera_resolution == 480       # rgb_array => render_resolution (:213-218)
    assert K.objlock_config_from_reference_kwargs(**{**objlock_kw, "render_mode": None}).camera_resolution == 128
    c2 = K.objlock_config_from_reference_kwargs(**objlock_kw, camera_profile="cockpit_fpv", camera_position_offset=(0.5, 0.0, 0.2),
                                                camera_angle_degrees=-10, camera_FOV_degrees=60, camera_resolution=(64, 64),
                                                duck_urdf_path="duck_vhacd.urdf", flight_mode=0)
    assert list(c2.camera_offset) == [0.5, 0.0, 0.2] and c2.camera_angle_deg == -10.0 and c2.camera_fov_deg == 60.0
    assert c2.camera_resolution == 64
    # duck_vision_use_deltas=False (:69-70, 163-165, 440-441): 9 x 3 history values without the 4 deltas -> 22 + 3 + 27 = 52
    c3 = K.objlock_config_from_reference_kwargs(**{**objlock_kw, "duck_vision_use_deltas": False})
    assert c3.duck_vision_no_deltas == 1 and c.duck_vision_no_deltas == 0
    from pyflyt_drone_amd import _lib
    import ctypes as C
    assert _lib.lib().fw_obs_dim(C.byref(c3)) == 52 and _lib.lib().fw_obs_dim(C.byref(c)) == 56
    for bad, match in ((dict(duck_vision_history_len=5), "duck_vision_history_len"),
                       (dict(camera_profile="chase"), "camera_profile"), (dict(flight_mode=-1), "flight_mode"),
                       (dict(render_mode="human"), "render mode"), (dict(camera_resolution=(64, 48)), "square")):
        with pytest.raises(ValueError, match=match):
            K.objlock_config_from_reference_kwargs(**{**objlock_kw, **bad})
    combined_kw = dict(
        sparse_reward=False, num_targets=8, goal_reach_distance=8, render_mode="rgb_array", angle_representation="euler",
        flight_dome_size=100.0, max_duration_seconds=120.0, agent_hz=30, use_egl=False, wind_config=K.TRAIN_COMBINED_WIND,
        num_obstacles=20, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0), obstacle_safe_distance_m=5.0,
        obstacle_avoid_reward_scale=1.0, obstacle_avoid_max_penalty=2.0,
        duck_camera_capture_interval_steps=6, duck_lock_hold_steps=10, duck_strike_distance_m=8, duck_strike_reward=200.0,
        duck_lock_step_reward=0.1, duck_approach_reward_scale=0.05, duck_switch_min_consecutive_seen=2,
        duck_switch_min_area=0.0005, duck_global_scaling=30.0)
    c = K.waypoint_objlock_config_from_reference_kwargs(context_length=2, **combined_kw)
    assert _same_config(c, K.train_waypoint_objlock_config())
    with pytest.raises(ValueError, match="flight_mode"):
        K.waypoint_objlock_config_from_reference_kwargs(**{**combined_kw, "flight_mode": 4})


def test_env_is_a_real_sb3_vecenv_with_gymnasium_spaces_when_those_import():
    """SB3 asserts isinstance(env, VecEnv) and isinstance(space, gymnasium.spaces.Box).  Neither package exists in this
    image, so two minimal stand-in modules (the classes SB3 / gymnasium export under those names, nothing more) are put
    on sys.modules and the package is re-imported: the env class must derive from the real VecEnv and Box must be the real
    Box; without them the in-tree mirrors are used."""
    import importlib
    import sys
    import types
    from pyflyt_drone_amd import spaces as S0, vec_env as V0
    assert not S0.HAVE_GYMNASIUM and not V0.HAVE_SB3 and S0.Box is S0.MirrorBox

    class GymBox:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, np.dtype(dtype)

    class SB3VecEnv:
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
            self.render_mode = self.get_attr("render_mode")[0]

    fake = {"gymnasium": types.ModuleType("gymnasium"), "gymnasium.spaces": types.ModuleType("gymnasium.spaces"),
            "stable_baselines3": types.ModuleType("stable_baselines3"), "stable_baselines3.common": types.ModuleType("stable_baselines3.common"),
            "stable_baselines3.common.vec_env": types.ModuleType("stable_baselines3.common.vec_env"),
            "stable_baselines3.common.vec_env.base_vec_env": types.ModuleType("stable_baselines3.common.vec_env.base_vec_env")}
    fake["gymnasium.spaces"].Box = GymBox
    fake["stable_baselines3.common.vec_env.base_vec_env"].VecEnv = SB3VecEnv
    try:
        sys.modules.update(fake)
        S1 = importlib.reload(S0)
        V1 = importlib.reload(V0)
        assert S1.HAVE_GYMNASIUM and S1.Box is GymBox and V1.HAVE_SB3
        assert issubclass(V1.FixedwingVecEnv, SB3VecEnv) and issubclass(V1.FixedwingObjLockVecEnv, SB3VecEnv)
        assert V1.Box is GymBox
    finally:
        for k in fake:
            sys.modules.pop(k, None)
        importlib.reload(S0); importlib.reload(V0)
    assert S0.Box is S0.MirrorBox and not V0.HAVE_SB3
    # the per-env attribute API answers from the object and its config (no GPU needed for this part)
    e = object.__new__(V0.FixedwingVecEnv)
    e.num_envs, e.cfg = 3, K.train_waypoints_v3_config()
    assert e.get_attr("render_mode") == [None] * 3 and e.get_attr("flight_dome_size", [0, 2]) == [100.0, 100.0]
    assert e.get_attr("num_targets", 1) == [8] and e.env_is_wrapped(object) == [False] * 3
    e.set_attr("note", "x"); assert e.get_attr("note") == ["x"] * 3
    with pytest.raises(AttributeError, match="device-side configuration"):
        e.set_attr("flight_dome_size", 50.0)
    with pytest.raises(AttributeError):
        e.get_attr("no_such_thing")
    with pytest.raises(AttributeError):
        e.env_method("no_such_method")
    assert e.env_method("get_attr", "num_targets", indices=[0]) == [[8, 8, 8]]


def test_box_space():
    b = P.Box(-1.0, 1.0, (4,), np.float64)
    assert b.shape == (4,) and b.dtype == np.float64
    assert b.contains(np.zeros(4)) and not b.contains(np.full(4, 2.0)) and not b.contains(np.zeros(3))
    s = b.sample()
    assert s.shape == (4,) and b.contains(s)
    o = P.Box(-np.inf, np.inf, (28,), np.float64)
    assert o.sample().shape == (28,)
    assert b == P.Box(-1.0, 1.0, (4,), np.float64) and b != o


def test_no_cpu_fallback_without_gpu():
    """On a box without a HIP device the product must refuse to run, not emulate."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.FixedwingWaypointsVecEnv(4)
    cfg = K.train_waypoints_v3_config()
    h = C.c_void_p()
    rc = _lib.lib().fw_create(C.byref(cfg), 4, 0, 0, 0, C.byref(h))
    assert rc == K.FW_EHIP and not h
    assert b"no HIP device" in _lib.lib().fw_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyflyt-drone_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "fw_oracle" not in text and "libfw_oracle" not in text and "fwo_" not in text, f


def test_isa_check_flags_spill_stores_ahead_of_an_exec_restore():
    """tools/check_isa.py (run on the device assembly of every build by _lib.build): a VGPR saved to an AGPR -- and reloaded
    later -- between the label of a join block and its `s_or_b64 exec` is saved for the lanes of the branch only.  The sample
    is the shape hipcc 7.2 produced in a dev build of the combined-task kernel (lane index garbage in 7 of 8 lanes, memory
    fault in the take-over prefetch); the same block with the store after the restore, a predicated accumulator update of an
    MFMA kernel and a store staged through AGPRs are not flagged."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "check_isa.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    bad = """
_Z4kernv:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\tds_write_b64 v56, v[38:39] offset:144
.LBB0_2:
\tv_accvgpr_write_b32 a16, v208
\ts_or_b64 exec, exec, s[0:1]
\tds_read_b32 v2, v233 offset:1544
\tv_accvgpr_read_b32 v208, a16
\ts_endpgm
.Lfunc_end0:
"""
    good = bad.replace("\tv_accvgpr_write_b32 a16, v208\n\ts_or_b64 exec, exec, s[0:1]", "\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a16, v208")
    mfma = bad.replace("\tds_read_b32", "\tv_mfma_f32_32x32x2_f32 a[16:31], v0, v1, a[16:31]\n\tds_read_b32")      # a16 is an accumulator there
    mfma_elsewhere = bad.replace("\tds_read_b32", "\tv_mfma_f32_32x32x2_f32 a[32:47], v0, v1, a[32:47]\n\tds_read_b32")   # a16 is still a spill slot
    staged = bad.replace("\tv_accvgpr_write_b32 a16, v208\n", "\tv_accvgpr_write_b32 a16, v208\n\tglobal_store_dwordx4 v[82:83], a[16:19], off offset:48\n")
    assert [h[1] for h in m.scan(bad)] == [".LBB0_2"]
    assert m.scan(good) == [] and m.scan(mfma) == [] and m.scan(staged) == []
    assert [h[1] for h in m.scan(mfma_elsewhere)] == [".LBB0_2"]      # an MFMA somewhere in the function does not switch the check off


def test_isa_check_wants_a_vmcnt_wait_between_the_partial_stores_and_the_ticket():
    """tools/check_isa.py, second guard: in a last-block-done kernel the write-through stores of the partial sums must be
    drained (`s_waitcnt vmcnt(0)`) before the workgroup barrier behind which the ticket atomic is taken -- otherwise the
    ticket can overtake the partials and the last block folds stale words."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "check_isa.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    good = """
_Z23fw_collect_stats_kernelIdEvN5fwsim9StatsArgsE:
\tglobal_store_dwordx2 v[0:1], v[2:3], off sc1
\ts_waitcnt vmcnt(0)
\ts_barrier
\tglobal_atomic_add v4, v5, v6, s[0:1] sc0
\ts_endpgm
.Lfunc_end0:
"""
    bad = good.replace("\ts_waitcnt vmcnt(0)\n", "")
    other = bad.replace("fw_collect_stats_kernel", "fw_something_else_kernel")
    assert m.scan_ticket(good) == []
    assert [h[1] for h in m.scan_ticket(bad)] == ["ticket"]
    assert m.scan_ticket(other) == []


def test_a_one_rank_process_group_counts_as_a_sharded_job_only_when_forced(monkeypatch):
    """rollout._dist(): FW_DIST_FORCE=1 is what lets a one-GPU box run the collectives of the sharded path (one rank over RCCL in the
    -m gpu tests; here over gloo).  Sums and gathers over one rank are the identity."""
    import socket
    import torch
    import torch.distributed as td
    from pyflyt_drone_amd import rollout as R
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    for k, v in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port)), ("FW_DIST_BACKEND", "gloo")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("FW_DIST_FORCE", raising=False)
    assert R.init_distributed_from_env() == (1, 0, 0) and not td.is_initialized()
    monkeypatch.setenv("FW_DIST_FORCE", "1")
    try:
        assert R.init_distributed_from_env() == (1, 0, 0) and td.is_initialized() and R._dist() is td
        x = torch.arange(6, dtype=torch.float64).reshape(2, 3)
        assert torch.equal(R.all_reduce_sum_(x.clone()), x) and torch.equal(R.all_gather_cat(x), x)
        monkeypatch.delenv("FW_DIST_FORCE")
        assert R._dist() is None
    finally:
        if td.is_initialized():
            td.destroy_process_group()
