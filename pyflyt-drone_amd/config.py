"""ctypes mirror of ``include/fwsim.h:fw_config`` and the build-owned defaults.

The aero/motor coefficients are the values of the reference's
``my_models/fixedwing/fixewing.yaml:1-71`` (the only physics data in the
reference).  Everything the reference does not contain -- mass, inertia, link
origins, collision geometry, the mode-0 mixer, the wind coupling -- is an
explicit field here (SURVEY.md section 7 "hard parts"): the build owns those
numbers and does not claim they equal PyFlyt's URDF.

Task parameter defaults follow the three training scripts:
``train/train_Fixedwing_Waypoints_v3.py:27-55,100-110``,
``train/train_objlock.py:27-86`` and
``train/train_Fixedwing_Waypoints_ObjLock.py:35-92``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Mapping, Optional

FW_ABI_VERSION = 7
FW_NUM_SURFACES = 5
FW_NUM_ACTUATORS = 6
FW_MAX_TARGETS = 8
FW_MAX_COLLISION_PTS = 8
FW_MAX_OBSTACLES = 20
FW_VISION_FEATS = 9
FW_VISION_HIST = 3
FW_STATE_DIM = 176
FW_INFO_DIM = 8

FW_OK, FW_EINVAL, FW_EHIP, FW_ENOMEM, FW_EVERSION, FW_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
FW_TASK_WAYPOINTS, FW_TASK_OBJLOCK, FW_TASK_WAYPOINT_OBJLOCK = 0, 1, 2
FW_F64, FW_F32 = 0, 1
FW_WIND_OFF, FW_WIND_CONSTANT, FW_WIND_GUST_SINE = 0, 1, 2
FW_WIND_COUPLE_NONE, FW_WIND_COUPLE_FORCE, FW_WIND_COUPLE_AIRSPEED = 0, 1, 2

# canonical state record offsets (fwsim.h FW_S_*)
S_POS, S_QUAT, S_VEL, S_OMEGA, S_ACT, S_ACTION = 0, 3, 7, 10, 13, 19
S_STEP_COUNT, S_TICK_COUNT, S_EPISODE, S_FLAGS, S_NUM_REACHED, S_NEW_DIST = 23, 24, 25, 26, 27, 28
S_WIND, S_EP_RETURN, S_TARGETS, S_TASK = 29, 36, 37, 61
# objlock tail (offsets from S_TASK)
ST_DUCK_POS, ST_LOCK_STEPS, ST_PREV_EST, ST_LAST_CX, ST_LAST_CY = 0, 3, 4, 5, 6
ST_LAST_AREA, ST_LAST_DEPTH, ST_SINCE_SEEN, ST_HIST_FILLED, ST_FRAME_HAS, ST_FRAME, ST_HIST = 7, 8, 9, 10, 11, 12, 20
ST_DUCK_PHASE, ST_SEEN_CONSEC, ST_NUM_OBST, ST_OBST, ST_DIM = 47, 48, 49, 50, 110

INFO_NUM_TARGETS_REACHED, INFO_COLLISION, INFO_OUT_OF_BOUNDS, INFO_ENV_COMPLETE = 0, 1, 2, 3
INFO_DUCK_STRIKE, INFO_IS_SUCCESS, INFO_EP_LEN = 4, 5, 6
# fw_get_counters columns (fwsim.h FW_CTR_*)
CTR_LAUNCHES, CTR_RESETS, CTR_SHADOW_HITS, CTR_SCENARIO_HITS, CTR_FALLBACKS, CTR_HELPER_TIMEOUTS, FW_CTR_DIM = 0, 1, 2, 3, 4, 5, 8


class SurfaceParams(C.Structure):
    _fields_ = [
        ("Cl_alpha_2D", C.c_double), ("chord", C.c_double), ("span", C.c_double),
        ("flap_to_chord", C.c_double), ("eta", C.c_double), ("alpha_0_base_deg", C.c_double),
        ("alpha_stall_P_base_deg", C.c_double), ("alpha_stall_N_base_deg", C.c_double),
        ("Cd_0", C.c_double), ("deflection_limit_deg", C.c_double), ("tau", C.c_double),
        ("lift_unit", C.c_double * 3), ("forward_unit", C.c_double * 3), ("pos", C.c_double * 3),
    ]


class MotorParams(C.Structure):
    _fields_ = [
        ("total_thrust", C.c_double), ("thrust_coef", C.c_double), ("torque_coef", C.c_double),
        ("noise_ratio", C.c_double), ("tau", C.c_double),
        ("thrust_unit", C.c_double * 3), ("pos", C.c_double * 3),
    ]


class FwConfig(C.Structure):
    _fields_ = [
        # ints
        ("abi_version", C.c_int32), ("task", C.c_int32), ("dtype", C.c_int32),
        ("angle_representation", C.c_int32), ("agent_hz", C.c_int32), ("physics_hz", C.c_int32),
        ("control_hz", C.c_int32), ("warmup_aviary_steps", C.c_int32), ("auto_reset", C.c_int32),
        ("sparse_reward", C.c_int32), ("num_targets", C.c_int32), ("context_length", C.c_int32),
        ("wind_mode", C.c_int32), ("wind_randomize_on_reset", C.c_int32),
        ("wind_randomize_phase", C.c_int32), ("wind_coupling", C.c_int32), ("gyroscopic", C.c_int32),
        ("n_collision_pts", C.c_int32), ("num_obstacles", C.c_int32),
        ("duck_camera_capture_interval_steps", C.c_int32), ("duck_lock_hold_steps", C.c_int32),
        ("duck_lock_decay_steps", C.c_int32), ("duck_switch_min_consecutive_seen", C.c_int32),
        ("camera_resolution", C.c_int32), ("duck_vision_no_deltas", C.c_int32), ("reserved_i", C.c_int32 * 7),
        # env / task scalars
        ("flight_dome_size", C.c_double), ("max_duration_seconds", C.c_double),
        ("goal_reach_distance", C.c_double), ("waypoint_min_height", C.c_double),
        ("waypoint_spawn_size", C.c_double),
        ("start_pos", C.c_double * 3), ("start_orn", C.c_double * 3), ("start_vel", C.c_double * 3),
        # wind
        ("wind_enu_mps", C.c_double * 3), ("wind_enu_mps_range", (C.c_double * 2) * 3),
        ("gust_amp_enu_mps", C.c_double * 3), ("gust_amp_enu_mps_range", (C.c_double * 2) * 3),
        ("gust_freq_hz", C.c_double), ("gust_phase_rad", C.c_double), ("wind_force_coef", C.c_double),
        # vehicle
        ("mass", C.c_double), ("inertia", C.c_double * 6), ("gravity", C.c_double),
        ("air_density", C.c_double),
        ("collision_pts", (C.c_double * 3) * FW_MAX_COLLISION_PTS),
        ("mixer", (C.c_double * 4) * FW_NUM_ACTUATORS),
        ("surfaces", SurfaceParams * FW_NUM_SURFACES), ("motor", MotorParams),
        # objlock
        ("duck_strike_distance_m", C.c_double), ("duck_strike_reward", C.c_double),
        ("duck_lock_step_reward", C.c_double), ("duck_approach_reward_scale", C.c_double),
        ("duck_global_scaling", C.c_double), ("duck_distance_reward_scale", C.c_double),
        ("duck_lock_center_radius", C.c_double), ("duck_centering_reward_scale", C.c_double),
        ("duck_visible_step_reward", C.c_double), ("duck_area_reward_scale", C.c_double),
        ("duck_lock_lost_penalty", C.c_double), ("duck_approach_reward_clip_m", C.c_double),
        ("duck_switch_min_area", C.c_double), ("duck_radius_per_scale", C.c_double),
        ("obstacle_radius", C.c_double), ("obstacle_height_range", C.c_double * 2),
        ("obstacle_safe_distance_m", C.c_double), ("obstacle_avoid_reward_scale", C.c_double),
        ("obstacle_avoid_max_penalty", C.c_double),
        ("camera_offset", C.c_double * 3), ("camera_angle_deg", C.c_double),
        ("camera_fov_deg", C.c_double), ("camera_near", C.c_double), ("camera_far", C.c_double),
        ("reserved_d", C.c_double * 8),
    ]

    def copy(self) -> "FwConfig":
        out = FwConfig()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(FwConfig))
        return out


# --- coefficient data: my_models/fixedwing/fixewing.yaml:8-71 (degrees as in the yaml) ---
_YAML_SURFACES = {
    "main_wing": dict(Cl_alpha_2D=6.283, chord=0.3, span=1.6, flap_to_chord=0.3, eta=0.65,
                      alpha_0_base=-2, alpha_stall_P_base=14, alpha_stall_N_base=-9, Cd_0=0.01,
                      deflection_limit=0, tau=0.05),
    "left_wing_flapped": dict(Cl_alpha_2D=6.283, chord=0.3, span=0.3, flap_to_chord=0.3, eta=0.65,
                              alpha_0_base=-2, alpha_stall_P_base=14, alpha_stall_N_base=-9, Cd_0=0.01,
                              deflection_limit=30, tau=0.05),
    "right_wing_flapped": dict(Cl_alpha_2D=6.283, chord=0.3, span=0.3, flap_to_chord=0.3, eta=0.65,
                               alpha_0_base=-2, alpha_stall_P_base=14, alpha_stall_N_base=-9, Cd_0=0.01,
                               deflection_limit=30, tau=0.05),
    "horizontal_tail": dict(Cl_alpha_2D=6.283, chord=0.2, span=0.625, flap_to_chord=0.3, eta=0.65,
                            alpha_0_base=0, alpha_stall_P_base=9, alpha_stall_N_base=-9, Cd_0=0.01,
                            deflection_limit=20, tau=0.05),
    "vertical_tail": dict(Cl_alpha_2D=6.283, chord=0.2, span=0.312, flap_to_chord=0.3, eta=0.65,
                          alpha_0_base=0, alpha_stall_P_base=9, alpha_stall_N_base=-9, Cd_0=0.01,
                          deflection_limit=20, tau=0.05),
}
# my_models/fixedwing/fixewing.yaml:1-6
_YAML_MOTOR = dict(total_thrust=18, thrust_coef=3.16e-10, torque_coef=7.94e-12, noise_ratio=0.02, tau=0.01)

# aux_state / actuator order (the reference author's own listing of mode -1:
# envs/fixedwing_envs/fixedwing_lowlevel_env.py:14 "[left_ail, right_ail, hstab, vstab, flap, thrust]")
SURFACE_ORDER = ("left_wing_flapped", "right_wing_flapped", "horizontal_tail", "vertical_tail", "main_wing")

# Build-owned geometry (body frame: x forward, y left, z up; origin = composite COM).
_GEOMETRY = {
    "left_wing_flapped": dict(lift=(0, 0, 1), fwd=(1, 0, 0), pos=(-0.03, 0.95, 0.03)),
    "right_wing_flapped": dict(lift=(0, 0, 1), fwd=(1, 0, 0), pos=(-0.03, -0.95, 0.03)),
    "horizontal_tail": dict(lift=(0, 0, 1), fwd=(1, 0, 0), pos=(-0.85, 0.0, 0.0)),
    "vertical_tail": dict(lift=(0, 1, 0), fwd=(1, 0, 0), pos=(-0.85, 0.0, 0.16)),
    "main_wing": dict(lift=(0, 0, 1), fwd=(1, 0, 0), pos=(-0.03, 0.0, 0.03)),
}
_MOTOR_GEOMETRY = dict(unit=(1, 0, 0), pos=(0.35, 0.0, 0.0))
_MASS = 2.0
_INERTIA = (0.22, 0.17, 0.36, 0.0, 0.0, 0.0)   # ixx iyy izz ixy ixz iyz
# nose, tail, wing tips, belly, fin top
_COLLISION_PTS = ((0.40, 0, 0), (-0.95, 0, 0), (0, 1.10, 0.03), (0, -1.10, 0.03), (0, 0, -0.10), (-0.85, 0, 0.32))
# mode 0: [roll, pitch, yaw, thrust] -> [left_ail, right_ail, h-tail, v-tail, main, throttle]
_MIXER = ((-1, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 0), (0, 0, 0, 1))


def _set_vec(dst, src):
    for i, v in enumerate(src):
        dst[i] = float(v)


def _fill_vehicle(c: FwConfig) -> None:
    c.mass = _MASS
    _set_vec(c.inertia, _INERTIA)
    c.gravity = 9.81
    c.air_density = 1.225
    c.n_collision_pts = len(_COLLISION_PTS)
    for i, p in enumerate(_COLLISION_PTS):
        _set_vec(c.collision_pts[i], p)
    for a, row in enumerate(_MIXER):
        _set_vec(c.mixer[a], row)
    for s, name in enumerate(SURFACE_ORDER):
        y, g, sp = _YAML_SURFACES[name], _GEOMETRY[name], c.surfaces[s]
        sp.Cl_alpha_2D, sp.chord, sp.span = y["Cl_alpha_2D"], y["chord"], y["span"]
        sp.flap_to_chord, sp.eta = y["flap_to_chord"], y["eta"]
        sp.alpha_0_base_deg = y["alpha_0_base"]
        sp.alpha_stall_P_base_deg = y["alpha_stall_P_base"]
        sp.alpha_stall_N_base_deg = y["alpha_stall_N_base"]
        sp.Cd_0, sp.deflection_limit_deg, sp.tau = y["Cd_0"], y["deflection_limit"], y["tau"]
        _set_vec(sp.lift_unit, g["lift"]); _set_vec(sp.forward_unit, g["fwd"]); _set_vec(sp.pos, g["pos"])
    m = c.motor
    m.total_thrust, m.thrust_coef, m.torque_coef = _YAML_MOTOR["total_thrust"], _YAML_MOTOR["thrust_coef"], _YAML_MOTOR["torque_coef"]
    m.noise_ratio, m.tau = _YAML_MOTOR["noise_ratio"], _YAML_MOTOR["tau"]
    _set_vec(m.thrust_unit, _MOTOR_GEOMETRY["unit"]); _set_vec(m.pos, _MOTOR_GEOMETRY["pos"])


_WIND_MODES = {"constant": FW_WIND_CONSTANT, "gust_sine": FW_WIND_GUST_SINE}
_WIND_COUPLING = {"none": FW_WIND_COUPLE_NONE, "force": FW_WIND_COUPLE_FORCE, "airspeed": FW_WIND_COUPLE_AIRSPEED}


def apply_wind_config(c: FwConfig, wind: Optional[Mapping[str, Any]]) -> None:
    """Translate the reference's wind dict (configs/env.yaml:20-30,
    envs/fixedwing_envs/fixedwing_base_env.py:108-173) into fw_config fields,
    raising ``ValueError`` for exactly the inputs the reference rejects."""
    cfg = dict(wind or {})
    if not bool(cfg.get("enabled", False)):
        c.wind_mode = FW_WIND_OFF
        return
    mode = str(cfg.get("mode", "constant")).lower()
    if mode not in _WIND_MODES:
        raise ValueError(f"Unsupported wind mode: {mode}")
    c.wind_mode = _WIND_MODES[mode]
    c.wind_randomize_on_reset = int(bool(cfg.get("randomize_on_reset", False)))
    c.wind_randomize_phase = int(bool(cfg.get("randomize_gust_phase", True)))
    coupling = str(cfg.get("coupling", "force")).lower()
    if coupling not in _WIND_COUPLING:
        raise ValueError(f"Unsupported wind coupling: {coupling}")
    c.wind_coupling = _WIND_COUPLING[coupling]
    c.wind_force_coef = float(cfg.get("force_coef", 1.0))

    def vec3(base_key, range_key, dst, dst_range):
        base = cfg.get(base_key, (0.0, 0.0, 0.0))
        if len(base) != 3:
            raise ValueError(f"Invalid {base_key}: {base}")
        _set_vec(dst, base)
        ranges = cfg.get(range_key, None)
        if ranges is None:
            for i in range(3):
                dst_range[i][0] = dst_range[i][1] = float(base[i])
            return
        if (not isinstance(ranges, (list, tuple)) or len(ranges) != 3
                or not all(isinstance(r, (list, tuple)) and len(r) == 2 for r in ranges)):
            raise ValueError(f"Invalid {range_key}: {ranges}")
        for i in range(3):
            dst_range[i][0], dst_range[i][1] = float(ranges[i][0]), float(ranges[i][1])

    vec3("wind_enu_mps", "wind_enu_mps_range", c.wind_enu_mps, c.wind_enu_mps_range)
    vec3("gust_amp_enu_mps", "gust_amp_enu_mps_range", c.gust_amp_enu_mps, c.gust_amp_enu_mps_range)
    c.gust_freq_hz = float(cfg.get("gust_freq_hz", 0.0))
    c.gust_phase_rad = float(cfg.get("gust_phase_rad", 0.0))


def _angle_repr(angle_representation: str) -> int:
    if angle_representation == "euler":
        return 0
    if angle_representation == "quaternion":
        return 1
    raise ValueError(
        f"angle_representation must be either `euler` or `quaternion`, not {angle_representation}")


class FwCollectArgs(C.Structure):
    """``fw_collect_args`` of include/fwsim.h (one-launch vec-step of the rollout collector)."""
    _fields_ = ([(n, C.c_void_p) for n in ("params", "obs_mean", "obs_var", "obs_count", "returns", "ret_mean", "ret_var", "ret_count",
                                           "obs_acc", "ret_acc", "rng", "obs_copy", "act_raw", "logp", "value", "act_env", "rew_out",
                                           "start_out", "obs", "reward", "terminated", "truncated", "terminal_obs", "info_i32", "workspace")]
                + [("workspace_bytes", C.c_int64), ("trace", C.c_void_p), ("gamma", C.c_double)]
                + [(n, C.c_float) for n in ("clip_obs", "eps_obs", "clip_reward", "eps_reward")]
                + [(n, C.c_int32) for n in ("update_obs", "update_ret", "norm_reward", "deterministic")])


class FwCollectCloseArgs(C.Structure):
    """``fw_collect_close_args`` of include/fwsim.h (GAE buffers of the rollout-closing launch)."""
    _fields_ = ([(n, C.c_void_p) for n in ("rewards", "values", "episode_starts", "adv", "ret")]
                + [("T", C.c_int32), ("gae_gamma", C.c_float), ("gae_lambda", C.c_float)])


class FwScenario(C.Structure):
    """``fw_scenario`` of include/fwsim.h: host arrays of a caller-supplied scenario (NULL = keep the env's own draw)."""
    _fields_ = [("targets", C.c_void_p), ("duck_pos", C.c_void_p), ("obstacles", C.c_void_p), ("num_obstacles", C.c_void_p),
                ("wind_base", C.c_void_p), ("gust_amp", C.c_void_p), ("gust_phase", C.c_void_p)]


def make_scenario(num_envs: int, *, targets=None, duck_pos=None, obstacles=None, num_obstacles=None, wind_base=None,
                  gust_amp=None, gust_phase=None):
    """Build an :class:`FwScenario` from array-likes ``targets[N, <=8, 3]``, ``duck_pos[N,3]``, ``obstacles[N, <=20, 3]``
    (x, y, height) with ``num_obstacles[N]``, ``wind_base[N,3]``, ``gust_amp[N,3]``, ``gust_phase[N]``.  Returns
    ``(scenario, keepalive)``: the structure points into the arrays of ``keepalive``."""
    import numpy as np
    n, keep, sc = int(num_envs), [], FwScenario()

    def put(field, arr, shape, dtype=np.float64):
        if arr is None:
            return
        a = np.asarray(arr, dtype=dtype)
        if a.ndim == len(shape) and a.shape[0] == n and len(shape) == 3 and a.shape[1] <= shape[1] and a.shape[2] == 3:
            full = np.zeros(shape, dtype=dtype); full[:, :a.shape[1]] = a; a = full
        if a.shape != shape:
            raise ValueError(f"scenario field {field}: expected shape {shape}, got {a.shape}")
        a = np.ascontiguousarray(a); keep.append(a)
        setattr(sc, field, a.ctypes.data)
    put("targets", targets, (n, FW_MAX_TARGETS, 3)); put("duck_pos", duck_pos, (n, 3))
    put("obstacles", obstacles, (n, FW_MAX_OBSTACLES, 3)); put("num_obstacles", num_obstacles, (n,), np.int32)
    put("wind_base", wind_base, (n, 3)); put("gust_amp", gust_amp, (n, 3)); put("gust_phase", gust_phase, (n,))
    if obstacles is not None and num_obstacles is None:
        raise ValueError("scenario: obstacles need num_obstacles")
    return sc, keep


def check_agent_hz(agent_hz: int) -> None:
    """envs/fixedwing_envs/fixedwing_base_env.py:48-53."""
    if agent_hz <= 0 or 120 % agent_hz != 0:
        lowest = int(120 / (int(120 / agent_hz) + 1))
        highest = int(120 / int(120 / agent_hz))
        raise ValueError(f"`agent_hz` must be round denominator of 120, try {lowest} or {highest}.")


def base_config(*, task: int, dtype: str = "float64", angle_representation: str = "quaternion",
                agent_hz: int = 30, flight_dome_size: float = 100.0, max_duration_seconds: float = 120.0,
                start_pos=(0.0, 0.0, 10.0), wind_config: Optional[Mapping[str, Any]] = None,
                motor_noise: bool = True, auto_reset: bool = True) -> FwConfig:
    check_agent_hz(int(agent_hz))
    c = FwConfig()
    c.abi_version = FW_ABI_VERSION
    c.task = task
    if dtype not in ("float64", "float32"):
        raise ValueError(f"dtype must be float64 or float32, not {dtype}")
    c.dtype = FW_F64 if dtype == "float64" else FW_F32
    c.angle_representation = _angle_repr(angle_representation)
    c.agent_hz = int(agent_hz)
    c.physics_hz, c.control_hz = 240, 120
    c.warmup_aviary_steps = 10
    c.auto_reset = int(auto_reset)
    c.gyroscopic = 1
    c.flight_dome_size = float(flight_dome_size)
    c.max_duration_seconds = float(max_duration_seconds)
    c.waypoint_min_height = 0.5
    c.waypoint_spawn_size = float(flight_dome_size)
    _set_vec(c.start_pos, start_pos)
    _set_vec(c.start_orn, (0.0, 0.0, 0.0))
    _set_vec(c.start_vel, (20.0, 0.0, 0.0))
    _fill_vehicle(c)
    if not motor_noise:
        c.motor.noise_ratio = 0.0
    c.wind_force_coef = 1.0
    apply_wind_config(c, wind_config)
    # camera defaults (cockpit_fpv, envs/fixedwing_objlock_env.py:184-231,692)
    _set_vec(c.camera_offset, (0.8, 0.0, 0.12))
    c.camera_angle_deg, c.camera_fov_deg = -5.0, 90.0
    c.camera_near, c.camera_far = 0.1, 255.0
    c.camera_resolution = 128
    return c


def waypoints_config(*, sparse_reward: bool = False, num_targets: int = 4, goal_reach_distance: float = 2.0,
                     flight_dome_size: float = 100.0, max_duration_seconds: float = 120.0,
                     angle_representation: str = "quaternion", agent_hz: int = 30, context_length: int = 2,
                     wind_config: Optional[Mapping[str, Any]] = None, dtype: str = "float64",
                     motor_noise: bool = True, auto_reset: bool = True) -> FwConfig:
    """``PyFlyt/Fixedwing-Waypoints-v3`` + ``FlattenWaypointEnv`` (keyword names and
    defaults follow the upstream constructor as used at
    train/train_Fixedwing_Waypoints_v3.py:100-117)."""
    if not 0 <= int(num_targets) <= FW_MAX_TARGETS:
        raise ValueError(f"num_targets must be in [0, {FW_MAX_TARGETS}]")
    c = base_config(task=FW_TASK_WAYPOINTS, dtype=dtype, angle_representation=angle_representation,
                    agent_hz=agent_hz, flight_dome_size=flight_dome_size,
                    max_duration_seconds=max_duration_seconds, start_pos=(0.0, 0.0, 10.0),
                    wind_config=wind_config, motor_noise=motor_noise, auto_reset=auto_reset)
    c.sparse_reward = int(bool(sparse_reward))
    c.num_targets = int(num_targets)
    c.goal_reach_distance = float(goal_reach_distance)
    c.context_length = int(context_length)
    return c


def train_waypoints_v3_config(**overrides) -> FwConfig:
    """TRAIN_CONFIG of train/train_Fixedwing_Waypoints_v3.py:27-55 (the headline config)."""
    kw = dict(sparse_reward=True, num_targets=8, goal_reach_distance=4.0, flight_dome_size=100.0,
              max_duration_seconds=120.0, angle_representation="euler", agent_hz=30, context_length=2,
              wind_config={"enabled": False, "mode": "constant", "wind_enu_mps": [0.0, 0.0, 0.0]})
    kw.update(overrides)
    return waypoints_config(**kw)


def _fill_objlock(c: FwConfig, *, num_obstacles, obstacle_radius, obstacle_height_range, obstacle_safe_distance_m,
                  obstacle_avoid_reward_scale, obstacle_avoid_max_penalty, duck_camera_capture_interval_steps,
                  duck_lock_hold_steps, duck_strike_distance_m, duck_strike_reward, duck_lock_step_reward,
                  duck_approach_reward_scale, duck_global_scaling, duck_distance_reward_scale=1.0,
                  duck_lock_center_radius=0.55, duck_centering_reward_scale=3.0, duck_visible_step_reward=2.0,
                  duck_area_reward_scale=5.0, duck_lock_decay_steps=1, duck_lock_lost_penalty=0.5,
                  duck_approach_reward_clip_m=2.0, camera_resolution=128) -> None:
    if not 0 <= int(num_obstacles) <= FW_MAX_OBSTACLES:
        raise ValueError(f"num_obstacles must be in [0, {FW_MAX_OBSTACLES}]")
    c.num_obstacles = int(num_obstacles)
    c.obstacle_radius = float(obstacle_radius)
    lo, hi = float(obstacle_height_range[0]), float(obstacle_height_range[1])
    if hi < lo:                                   # envs/fixedwing_objlock_env.py:525-526
        lo, hi = hi, lo
    c.obstacle_height_range[0], c.obstacle_height_range[1] = lo, hi
    c.obstacle_safe_distance_m = float(obstacle_safe_distance_m)
    c.obstacle_avoid_reward_scale = float(obstacle_avoid_reward_scale)
    c.obstacle_avoid_max_penalty = float(obstacle_avoid_max_penalty)
    c.duck_camera_capture_interval_steps = int(duck_camera_capture_interval_steps)
    c.duck_lock_hold_steps = int(duck_lock_hold_steps)
    c.duck_lock_decay_steps = int(max(1, duck_lock_decay_steps))            # :116
    c.duck_strike_distance_m = float(duck_strike_distance_m)
    c.duck_strike_reward = float(duck_strike_reward)
    c.duck_lock_step_reward = float(duck_lock_step_reward)
    c.duck_approach_reward_scale = float(duck_approach_reward_scale)
    c.duck_global_scaling = float(duck_global_scaling)
    c.duck_distance_reward_scale = float(duck_distance_reward_scale)
    c.duck_lock_center_radius = float(duck_lock_center_radius)
    c.duck_centering_reward_scale = float(duck_centering_reward_scale)
    c.duck_visible_step_reward = float(duck_visible_step_reward)
    c.duck_area_reward_scale = float(duck_area_reward_scale)
    c.duck_lock_lost_penalty = float(duck_lock_lost_penalty)
    c.duck_approach_reward_clip_m = float(max(0.0, duck_approach_reward_clip_m))   # :118
    c.duck_radius_per_scale = 0.05           # build-owned: analytic duck = sphere of radius scale * 0.05 m
    c.camera_resolution = int(camera_resolution)


def objlock_config(*, sparse_reward: bool = False, flight_dome_size: float = 100.0, max_duration_seconds: float = 120.0,
                   angle_representation: str = "quaternion", agent_hz: int = 30,
                   num_obstacles: int = 5, obstacle_radius: float = 2.0, obstacle_height_range=(10.0, 30.0),
                   obstacle_safe_distance_m: float = 20.0, obstacle_avoid_reward_scale: float = 1.0,
                   obstacle_avoid_max_penalty: float = 2.0, duck_camera_capture_interval_steps: int = 12,
                   duck_lock_hold_steps: int = 10, duck_strike_distance_m: float = 2.0, duck_strike_reward: float = 200.0,
                   duck_lock_step_reward: float = 0.1, duck_approach_reward_scale: float = 0.05,
                   duck_global_scaling: float = 20.0, duck_distance_reward_scale: float = 1.0,
                   duck_lock_center_radius: float = 0.55, duck_centering_reward_scale: float = 3.0,
                   duck_visible_step_reward: float = 2.0, duck_area_reward_scale: float = 5.0,
                   duck_lock_decay_steps: int = 1, duck_lock_lost_penalty: float = 0.5,
                   duck_approach_reward_clip_m: float = 2.0, camera_resolution: int = 128,
                   wind_config: Optional[Mapping[str, Any]] = None, dtype: str = "float64",
                   motor_noise: bool = True, auto_reset: bool = True) -> FwConfig:
    """``FixedwingObjLockEnv`` + ``FlattenObjLockEnv``: keyword names and defaults of
    envs/fixedwing_objlock_env.py:37-81 (start position [0,0,100] :83)."""
    c = base_config(task=FW_TASK_OBJLOCK, dtype=dtype, angle_representation=angle_representation, agent_hz=agent_hz,
                    flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds,
                    start_pos=(0.0, 0.0, 100.0), wind_config=wind_config, motor_noise=motor_noise, auto_reset=auto_reset)
    c.sparse_reward = int(bool(sparse_reward))
    c.num_targets = 0
    c.context_length = 0
    _fill_objlock(c, num_obstacles=num_obstacles, obstacle_radius=obstacle_radius,
                  obstacle_height_range=obstacle_height_range, obstacle_safe_distance_m=obstacle_safe_distance_m,
                  obstacle_avoid_reward_scale=obstacle_avoid_reward_scale,
                  obstacle_avoid_max_penalty=obstacle_avoid_max_penalty,
                  duck_camera_capture_interval_steps=duck_camera_capture_interval_steps,
                  duck_lock_hold_steps=duck_lock_hold_steps, duck_strike_distance_m=duck_strike_distance_m,
                  duck_strike_reward=duck_strike_reward, duck_lock_step_reward=duck_lock_step_reward,
                  duck_approach_reward_scale=duck_approach_reward_scale, duck_global_scaling=duck_global_scaling,
                  duck_distance_reward_scale=duck_distance_reward_scale, duck_lock_center_radius=duck_lock_center_radius,
                  duck_centering_reward_scale=duck_centering_reward_scale,
                  duck_visible_step_reward=duck_visible_step_reward, duck_area_reward_scale=duck_area_reward_scale,
                  duck_lock_decay_steps=duck_lock_decay_steps, duck_lock_lost_penalty=duck_lock_lost_penalty,
                  duck_approach_reward_clip_m=duck_approach_reward_clip_m, camera_resolution=camera_resolution)
    return c


TRAIN_OBJLOCK_WIND = {      # train/train_objlock.py:74-85
    "enabled": True, "mode": "gust_sine", "wind_enu_mps": [0.0, 0.0, 0.0], "gust_amp_enu_mps": [0.0, 0.0, 0.0],
    "gust_freq_hz": 0.2, "gust_phase_rad": 0.0, "randomize_on_reset": True, "randomize_gust_phase": True,
    "wind_enu_mps_range": [[-10.0, 10.0], [-10.0, 10.0], [-0.1, 0.1]],
    "gust_amp_enu_mps_range": [[0.0, 3.0], [0.0, 3.0], [0.0, 0.3]],
}


def train_objlock_config(**overrides) -> FwConfig:
    """ENV_CONFIG of train/train_objlock.py:27-86,113-153 (config 3 of BASELINE.json):
    dome 200 m, 60 s, euler, duck scale 60, hold 5, strike 10 m / +400, lock step 0.2,
    approach 0.1, capture interval 12, no obstacles, gust wind; render_mode="rgb_array"
    makes the camera 480 x 480 (envs/fixedwing_objlock_env.py:213-218)."""
    kw = dict(sparse_reward=False, flight_dome_size=200.0, max_duration_seconds=60.0, angle_representation="euler",
              agent_hz=30, num_obstacles=0, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0),
              obstacle_safe_distance_m=10.0, obstacle_avoid_reward_scale=1.0, obstacle_avoid_max_penalty=5.0,
              duck_camera_capture_interval_steps=12, duck_lock_hold_steps=5, duck_strike_distance_m=10.0,
              duck_strike_reward=400.0, duck_lock_step_reward=0.2, duck_approach_reward_scale=0.1,
              duck_global_scaling=60.0, camera_resolution=480, wind_config=TRAIN_OBJLOCK_WIND)
    kw.update(overrides)
    return objlock_config(**kw)


def waypoint_objlock_config(*, sparse_reward: bool = False, num_targets: int = 4, goal_reach_distance: float = 2.0,
                            flight_dome_size: float = 100.0, max_duration_seconds: float = 120.0,
                            angle_representation: str = "quaternion", agent_hz: int = 30, context_length: int = 2,
                            num_obstacles: int = 5, obstacle_radius: float = 2.0, obstacle_height_range=(10.0, 30.0),
                            obstacle_safe_distance_m: float = 20.0, obstacle_avoid_reward_scale: float = 1.0,
                            obstacle_avoid_max_penalty: float = 2.0, duck_camera_capture_interval_steps: int = 6,
                            duck_lock_hold_steps: int = 10, duck_strike_distance_m: float = 2.0,
                            duck_strike_reward: float = 200.0, duck_lock_step_reward: float = 0.1,
                            duck_approach_reward_scale: float = 0.05, duck_switch_min_consecutive_seen: int = 2,
                            duck_switch_min_area: float = 0.0005, duck_global_scaling: float = 20.0,
                            waypoint_spawn_size: Optional[float] = None, camera_resolution: int = 128,
                            wind_config: Optional[Mapping[str, Any]] = None, dtype: str = "float64",
                            motor_noise: bool = True, auto_reset: bool = True) -> FwConfig:
    """``FlattenWaypointEnv(FixedwingWaypointObjLockEnv(...))``: keyword names and defaults of
    envs/fixedwing_waypoint_objlock_env.py:42-76 (start position [0,0,10] :78)."""
    if not 0 <= int(num_targets) <= FW_MAX_TARGETS:
        raise ValueError(f"num_targets must be in [0, {FW_MAX_TARGETS}]")
    c = base_config(task=FW_TASK_WAYPOINT_OBJLOCK, dtype=dtype, angle_representation=angle_representation,
                    agent_hz=agent_hz, flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds,
                    start_pos=(0.0, 0.0, 10.0), wind_config=wind_config, motor_noise=motor_noise, auto_reset=auto_reset)
    c.sparse_reward = int(bool(sparse_reward))
    c.num_targets = int(num_targets)
    c.goal_reach_distance = float(goal_reach_distance)
    c.context_length = int(context_length)
    c.waypoint_spawn_size = float(flight_dome_size if waypoint_spawn_size is None else waypoint_spawn_size)   # :94
    _fill_objlock(c, num_obstacles=num_obstacles, obstacle_radius=obstacle_radius,
                  obstacle_height_range=obstacle_height_range, obstacle_safe_distance_m=obstacle_safe_distance_m,
                  obstacle_avoid_reward_scale=obstacle_avoid_reward_scale,
                  obstacle_avoid_max_penalty=obstacle_avoid_max_penalty,
                  duck_camera_capture_interval_steps=duck_camera_capture_interval_steps,
                  duck_lock_hold_steps=duck_lock_hold_steps, duck_strike_distance_m=duck_strike_distance_m,
                  duck_strike_reward=duck_strike_reward, duck_lock_step_reward=duck_lock_step_reward,
                  duck_approach_reward_scale=duck_approach_reward_scale, duck_global_scaling=duck_global_scaling,
                  camera_resolution=camera_resolution)
    c.duck_switch_min_consecutive_seen = int(duck_switch_min_consecutive_seen)
    c.duck_switch_min_area = float(duck_switch_min_area)
    return c


TRAIN_COMBINED_WIND = {     # train/train_Fixedwing_Waypoints_ObjLock.py:59-70
    "enabled": True, "mode": "gust_sine", "wind_enu_mps": [0.0, 0.0, 0.0], "gust_amp_enu_mps": [0.0, 0.0, 0.0],
    "gust_freq_hz": 0.2, "gust_phase_rad": 0.0, "randomize_on_reset": True, "randomize_gust_phase": True,
    "wind_enu_mps_range": [[-5.0, 5.0], [-5.0, 5.0], [-0.5, 0.5]],
    "gust_amp_enu_mps_range": [[0.0, 3.0], [0.0, 3.0], [0.0, 0.3]],
}


def train_waypoint_objlock_config(**overrides) -> FwConfig:
    """TRAIN_CONFIG of train/train_Fixedwing_Waypoints_ObjLock.py:35-92,119-165 (config 5 of BASELINE.json)."""
    kw = dict(sparse_reward=False, num_targets=8, goal_reach_distance=8.0, flight_dome_size=100.0,
              max_duration_seconds=120.0, angle_representation="euler", agent_hz=30, context_length=2,
              num_obstacles=20, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0), obstacle_safe_distance_m=5.0,
              obstacle_avoid_reward_scale=1.0, obstacle_avoid_max_penalty=2.0, duck_camera_capture_interval_steps=6,
              duck_lock_hold_steps=10, duck_strike_distance_m=8.0, duck_strike_reward=200.0, duck_lock_step_reward=0.1,
              duck_approach_reward_scale=0.05, duck_switch_min_consecutive_seen=2, duck_switch_min_area=0.0005,
              duck_global_scaling=30.0, camera_resolution=480, wind_config=TRAIN_COMBINED_WIND)
    kw.update(overrides)
    return waypoint_objlock_config(**kw)


# ---------------------------------------------------------------------------------------------
# the reference constructors' full keyword sets (so that the reference's make_env() call sites work unchanged)
# ---------------------------------------------------------------------------------------------
def _camera_resolution_of(render_mode, render_resolution, camera_resolution) -> int:
    """envs/fixedwing_objlock_env.py:213-218: explicit camera_resolution, else render_resolution when a render mode is
    set, else (128, 128).  The analytic camera is square."""
    res = camera_resolution if camera_resolution is not None else (render_resolution if render_mode is not None else (128, 128))
    if isinstance(res, (int, float)):
        res = (int(res), int(res))
    w, h = int(res[0]), int(res[1])
    if w != h or w <= 0:
        raise ValueError(f"camera resolution must be square and positive on the device env, got {(w, h)}")
    return w


def _check_common_reference_kwargs(render_mode, flight_mode) -> None:
    if render_mode not in (None, "rgb_array"):
        raise ValueError(f"Invalid render mode {render_mode}, only [None, 'rgb_array'] have a device counterpart.")
    if int(flight_mode) != 0:
        raise ValueError(f"flight_mode {flight_mode} is not available on the device env (mode 0: [roll, pitch, yaw, thrust])")


def objlock_config_from_reference_kwargs(*, dtype: str = "float64", motor_noise: bool = True, auto_reset: bool = True,
                                         flight_mode: int = 0, render_mode=None, render_resolution=(480, 480),
                                         duck_urdf_path=None, use_egl: bool = False, camera_profile: str = "cockpit_fpv",
                                         camera_position_offset=None, camera_angle_degrees=None, camera_FOV_degrees=None,
                                         camera_resolution=None, duck_vision_history_len: int = 3,
                                         duck_vision_use_deltas: bool = True, **env_kwargs) -> FwConfig:
    """Every keyword of ``FixedwingObjLockEnv.__init__`` (envs/fixedwing_objlock_env.py:37-81), as passed by
    train/train_objlock.py:113-153.  Render-only arguments (``use_egl``, ``duck_urdf_path``) are accepted and ignored;
    what the device env cannot honour raises ``ValueError`` instead of being silently dropped."""
    _check_common_reference_kwargs(render_mode, flight_mode)
    del duck_urdf_path, use_egl                                   # renderer plumbing: no device counterpart needed
    if int(max(1, duck_vision_history_len)) != FW_VISION_HIST:
        raise ValueError(f"duck_vision_history_len must be {FW_VISION_HIST} on the device env (the observation layout is compiled in)")
    if camera_profile != "cockpit_fpv":
        raise ValueError(f"camera_profile {camera_profile!r} is not available on the device env (body-fixed 'cockpit_fpv' only; "
                         "'chase' is a tracking camera)")
    res = _camera_resolution_of(render_mode, render_resolution, camera_resolution)
    c = objlock_config(dtype=dtype, motor_noise=motor_noise, auto_reset=auto_reset, camera_resolution=res, **env_kwargs)
    c.duck_vision_no_deltas = 0 if bool(duck_vision_use_deltas) else 1      # :69-70, 440-441: history only, 52 values instead of 56
    if camera_position_offset is not None:                        # :194-201
        _set_vec(c.camera_offset, [float(v) for v in camera_position_offset])
    if camera_angle_degrees is not None:                          # :203-209 (the reference truncates to int)
        c.camera_angle_deg = float(int(camera_angle_degrees))
    if camera_FOV_degrees is not None:                            # :211-212
        c.camera_fov_deg = float(int(camera_FOV_degrees))
    return c


def waypoint_objlock_config_from_reference_kwargs(*, dtype: str = "float64", motor_noise: bool = True, auto_reset: bool = True,
                                                  context_length: int = 2, flight_mode: int = 0, render_mode=None,
                                                  render_resolution=(480, 480), duck_urdf_path=None, use_egl: bool = False,
                                                  **env_kwargs) -> FwConfig:
    """Every keyword of ``FixedwingWaypointObjLockEnv.__init__`` (envs/fixedwing_waypoint_objlock_env.py:42-76) plus the
    wrapper's ``context_length``, as passed by train/train_Fixedwing_Waypoints_ObjLock.py:119-165."""
    _check_common_reference_kwargs(render_mode, flight_mode)
    del duck_urdf_path, use_egl
    res = _camera_resolution_of(render_mode, render_resolution, env_kwargs.pop("camera_resolution", None))
    return waypoint_objlock_config(dtype=dtype, motor_noise=motor_noise, auto_reset=auto_reset, context_length=context_length,
                                   camera_resolution=res, **env_kwargs)


def obs_dim(c: FwConfig) -> int:
    att = (12 if c.angle_representation == 0 else 13) + 4 + 6
    if c.task == FW_TASK_OBJLOCK:
        return att + 3 + FW_VISION_FEATS * FW_VISION_HIST + (0 if c.duck_vision_no_deltas else 4)      # (:163-165)
    return att + 3 * c.context_length


def max_steps(c: FwConfig) -> int:
    """envs/fixedwing_envs/fixedwing_base_env.py:101."""
    return int(c.agent_hz * c.max_duration_seconds)


def env_step_ratio(c: FwConfig) -> int:
    """envs/fixedwing_envs/fixedwing_base_env.py:102."""
    return int(120 / c.agent_hz)


def max_rpm(c: FwConfig) -> float:
    return math.sqrt(c.motor.total_thrust / c.motor.thrust_coef)
