"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on identical (seed, action) traces.

Tolerances: BASELINE.json's north_star asks for 1e-4 on position/attitude; the
fp64 kernel is held to 1e-7 absolute on every observation / reward / state
element over multi-hundred-step traces with auto-resets (measured: ~1e-12), and
flags / info / episode boundaries must agree exactly.  The fp32 kernel is a
throughput mode and is characterised separately (single-step error, invariants).
"""
import ctypes as C

import numpy as np
import pytest

import pyflyt_drone_amd as P
from pyflyt_drone_amd import _lib
from pyflyt_drone_amd import config as K
from helpers import run_lockstep, seeded_actions

pytestmark = pytest.mark.gpu

GUST_FORCE = dict(enabled=True, mode="gust_sine", randomize_on_reset=True, randomize_gust_phase=True,
                  wind_enu_mps_range=[[-10, 10], [-10, 10], [-0.1, 0.1]],
                  gust_amp_enu_mps_range=[[0, 3], [0, 3], [0, 0.3]], gust_freq_hz=0.2)     # train/train_objlock.py:74-85
CONST_AIRSPEED = dict(enabled=True, mode="constant", wind_enu_mps=[2.0, -3.0, 0.25], coupling="airspeed")
CONST_RANDOM = dict(enabled=True, mode="constant", randomize_on_reset=True,
                    wind_enu_mps_range=[[-5, 5], [-5, 5], [-0.5, 0.5]])


def lockstep_cases():
    yield "train_v3", K.train_waypoints_v3_config(), "uniform"
    yield "train_v3_nonoise", K.train_waypoints_v3_config(motor_noise=False), "uniform"
    yield "dense_quat_ctx3_gust", K.waypoints_config(sparse_reward=False, num_targets=3, goal_reach_distance=8.0,
                                                      angle_representation="quaternion", context_length=3,
                                                      wind_config=GUST_FORCE), "gentle"
    yield "dense_big_reach", K.waypoints_config(sparse_reward=False, num_targets=8, goal_reach_distance=30.0,
                                                angle_representation="euler", context_length=2), "gentle"
    yield "sparse_big_reach_ctx1", K.waypoints_config(sparse_reward=True, num_targets=2, goal_reach_distance=40.0,
                                                      angle_representation="euler", context_length=1), "gentle"
    yield "const_airspeed", K.waypoints_config(sparse_reward=False, num_targets=4, goal_reach_distance=6.0,
                                               angle_representation="euler", wind_config=CONST_AIRSPEED,
                                               motor_noise=False), "uniform"
    yield "const_random_force", K.waypoints_config(sparse_reward=False, num_targets=4, goal_reach_distance=10.0,
                                                   angle_representation="euler", wind_config=CONST_RANDOM), "gentle"
    yield "short_episodes_trunc", K.waypoints_config(sparse_reward=True, num_targets=4, max_duration_seconds=1.0,
                                                     angle_representation="euler", flight_dome_size=1e4), "gentle"
    yield "no_gyro_agent60", _mutate(K.waypoints_config(sparse_reward=False, agent_hz=60, angle_representation="euler"),
                                      gyroscopic=0), "uniform"
    yield "no_autoreset", K.waypoints_config(sparse_reward=False, num_targets=2, goal_reach_distance=20.0,
                                             angle_representation="euler", auto_reset=False), "uniform"
    # ---- ObjLock task (envs/fixedwing_objlock_env.py), analytic camera
    yield "objlock_train", K.train_objlock_config(), "gentle"
    yield "objlock_train_uniform", K.train_objlock_config(), "uniform"
    yield "objlock_obstacles_quat", K.objlock_config(flight_dome_size=150.0, max_duration_seconds=20.0, num_obstacles=20,
                                                     obstacle_safe_distance_m=40.0, duck_camera_capture_interval_steps=2,
                                                     angle_representation="quaternion", camera_resolution=128,
                                                     wind_config=CONST_AIRSPEED), "gentle"
    # ---- combined task (envs/fixedwing_waypoint_objlock_env.py)
    yield "combined_train", K.train_waypoint_objlock_config(), "gentle"
    yield "combined_big_reach", K.waypoint_objlock_config(num_targets=3, goal_reach_distance=40.0, angle_representation="euler",
                                                          duck_camera_capture_interval_steps=1, num_obstacles=8,
                                                          obstacle_safe_distance_m=50.0, duck_strike_distance_m=30.0,
                                                          duck_lock_hold_steps=3, duck_global_scaling=60.0,
                                                          wind_config=CONST_RANDOM), "gentle"
    yield "combined_sparse_quat_ctx3", K.waypoint_objlock_config(sparse_reward=True, num_targets=2, goal_reach_distance=25.0,
                                                                  angle_representation="quaternion", context_length=3,
                                                                  duck_camera_capture_interval_steps=2, num_obstacles=0,
                                                                  motor_noise=False), "uniform"
    yield "objlock_sparse_nowind", K.objlock_config(sparse_reward=True, flight_dome_size=120.0, num_obstacles=3,
                                                    duck_camera_capture_interval_steps=1, angle_representation="euler",
                                                    motor_noise=False), "uniform"
    # duck_vision_use_deltas=False (envs/fixedwing_objlock_env.py:69-70, 440-441): the observation ends with the history, 52 wide
    yield "objlock_no_deltas", _mutate(K.train_objlock_config(duck_camera_capture_interval_steps=2), duck_vision_no_deltas=1), "gentle"


def _mutate(cfg, **kw):
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


CASES = {name: (cfg, kind) for name, cfg, kind in lockstep_cases()}


@pytest.fixture(params=[1, 8], ids=["lane_per_env", "8_lanes_per_env"])
def lanes(request, monkeypatch):
    """Both lane mappings of the kernel (fwsim_device.hpp) must pass every parity test."""
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", str(request.param))
    monkeypatch.setenv("FWSIM_G8_WAVES", "1")
    return request.param


@pytest.fixture(params=[1, 8, "8w2"], ids=["lane_per_env", "8_lanes_per_env", "8_lanes_2_waves_per_simd"])
def lanes3(request, monkeypatch):
    """... and, for the waypoints kernels, the 8-lane mapping built for two waves per SIMD (256 registers, wave-uniform tick
    constants): what fw_create picks between 8 192 and 16 384 envs."""
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8" if request.param == "8w2" else str(request.param))
    monkeypatch.setenv("FWSIM_G8_WAVES", "2" if request.param == "8w2" else "1")
    return request.param


@pytest.mark.parametrize("name", sorted(CASES))
def test_lockstep_f64(oracle, name, lanes3):
    cfg, kind = CASES[name]
    if lanes3 == "8w2" and cfg.task != K.FW_TASK_WAYPOINTS:
        pytest.skip("the two-waves-per-SIMD build exists for the waypoints kernels only")
    n = 192 + 7                                      # deliberately not a multiple of 64
    hip = P.FixedwingVecEnv(cfg, n, seed=1234)
    ora = oracle.OracleEnv(cfg, n, seed=1234)
    obj = cfg.task == K.FW_TASK_OBJLOCK
    assert name.split('_')[0] in ('objlock', 'combined') or cfg.task == K.FW_TASK_WAYPOINTS
    # ObjLock observations are float32-rounded (flatten_objlock_env.py:46): a 1e-13 difference can flip one f32 ulp
    worst = run_lockstep(hip, ora, 240, np.random.default_rng(5), kind=kind, atol=2e-5 if obj else 1e-7, rtol=0,
                         state_atol=1e-7)
    assert worst["obs"] < (2e-5 if obj else 1e-7) and worst["state"] < 1e-7
    if name in ("train_v3", "short_episodes_trunc", "dense_big_reach"):
        assert worst["dones"] > 0, "case was meant to exercise the auto-reset path"


def test_objlock_aimed_flights_lock_and_strike(oracle, lanes):
    """Aircraft aimed at their duck from 60-200 m: frames become visible, the lock counter runs,
    strikes (+400, is_success) happen -- identically in the kernel and the oracle."""
    _aimed_flights(oracle)


def test_objlock_aimed_flights_lock_and_strike_with_a_capture_wave(oracle, monkeypatch):
    """... and with the captures on a second wave (FWSIM_CAPTURE_WAVE=1): a strike is the one frame-dependent way a sub-step ends
    the agent step, i.e. the case in which the step wave must NOT run ahead of its frame."""
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8"); monkeypatch.setenv("FWSIM_CAPTURE_WAVE", "1")
    _aimed_flights(oracle, capture_wave=True)


def _aimed_flights(oracle, capture_wave=False):
    import torch
    cfg = K.train_objlock_config(duck_camera_capture_interval_steps=2, wind_config=None)
    n = 256
    hip = P.FixedwingVecEnv(cfg, n, seed=77); ora = oracle.OracleEnv(cfg, n, seed=77)
    assert hip.capture_wave == capture_wave
    hip.reset_tensor(); ora.reset()
    s = ora.get_state()
    rng = np.random.default_rng(8)
    T0 = K.S_TASK
    for i in range(n):
        rng_d = rng.uniform(60.0, 200.0); yaw = rng.uniform(-np.pi, np.pi)
        height = rng.uniform(8.0, 30.0)
        duck = s[i, T0:T0 + 3]
        # glide slope towards the duck: nose down by atan(height / range)  (positive pitch = nose down)
        pitch = np.arctan2(height, rng_d) * rng.uniform(0.6, 1.1)
        s[i, K.S_POS:K.S_POS + 3] = [duck[0] - rng_d * np.cos(yaw), duck[1] - rng_d * np.sin(yaw), height]
        s[i, K.S_QUAT:K.S_QUAT + 4] = oracle.quat_from_euler([0.0, pitch, yaw])
        v = oracle.mat_from_quat(s[i, K.S_QUAT:K.S_QUAT + 4]) @ np.array([20.0, 0, 0])
        s[i, K.S_VEL:K.S_VEL + 3] = v; s[i, K.S_OMEGA:K.S_OMEGA + 3] = 0
    hip.set_state(s); ora.set_state(s)
    np.testing.assert_allclose(hip.observe_tensor().cpu().numpy(), ora.observe(), rtol=0, atol=1e-6)
    strikes = visible = 0
    for t in range(150):
        a = seeded_actions(rng, n, "gentle") * 0.3
        o_obs, o_rew, o_term, o_trunc, o_tobs, o_info = ora.step(a)
        hip.step_tensor(torch.as_tensor(a, device=hip.device))
        assert np.array_equal(hip.terminated.cpu().numpy(), o_term) and np.array_equal(hip.truncated.cpu().numpy(), o_trunc), t
        assert np.array_equal(hip.info.cpu().numpy(), o_info), t
        np.testing.assert_allclose(hip.obs.cpu().numpy(), o_obs, rtol=0, atol=2e-5, err_msg=f"obs {t}")   # float32-rounded obs
        np.testing.assert_allclose(hip.rewards.cpu().numpy(), o_rew, rtol=0, atol=1e-6, err_msg=f"reward {t}")
        strikes += int(o_info[:, K.INFO_DUCK_STRIKE].sum()); visible += int((o_obs[:, 25] > 0.5).sum())
    assert visible > 500 and strikes >= 3, (visible, strikes)
    np.testing.assert_allclose(hip.get_state(), ora.get_state(), rtol=0, atol=1e-6)
    assert hip.get_counters()["capture_wave_timeouts"] == 0


def test_baseline_size_4096_envs_against_oracle(oracle, lanes3):
    """configs[1] of BASELINE.json: 4096 envs; 24 steps is what the scalar oracle does in ~2 s."""
    cfg = K.train_waypoints_v3_config()
    hip = P.FixedwingVecEnv(cfg, 4096, seed=42)
    ora = oracle.OracleEnv(cfg, 4096, seed=42)
    run_lockstep(hip, ora, 24, np.random.default_rng(0), kind="uniform", atol=1e-7, rtol=0)


@pytest.mark.parametrize("name", sorted(n for n in CASES if n.split("_")[0] in ("objlock", "combined")))
def test_lockstep_f64_with_a_capture_wave(oracle, name, monkeypatch):
    """The opt-in two-wave form of the camera step kernels (FWSIM_CAPTURE_WAVE=1; csrc/fwsim_objlock.hpp "The capture wave"): the
    captures run on a second wave of the workgroup one sub-step behind the physics, the frame-dependent half of the task logic
    follows them, the capture wave is the tile's shadow worker as well.  Same traces as the one-wave kernel against the oracle,
    resets and strikes included, and no wait between the two waves ever gave up."""
    cfg, kind = CASES[name]
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8"); monkeypatch.setenv("FWSIM_CAPTURE_WAVE", "1")
    n = 192 + 7
    hip = P.FixedwingVecEnv(cfg, n, seed=1234)
    assert hip.lanes_per_env == 8 and hip.capture_wave
    ora = oracle.OracleEnv(cfg, n, seed=1234)
    obj = cfg.task == K.FW_TASK_OBJLOCK
    worst = run_lockstep(hip, ora, 240, np.random.default_rng(5), kind=kind, atol=2e-5 if obj else 1e-7, rtol=0, state_atol=1e-7)
    assert worst["obs"] < (2e-5 if obj else 1e-7) and worst["state"] < 1e-7
    assert hip.get_counters()["capture_wave_timeouts"] == 0
    hip = P.FixedwingVecEnv(cfg, 4096, seed=42)
    ora = oracle.OracleEnv(cfg, 4096, seed=42)
    run_lockstep(hip, ora, 24, np.random.default_rng(0), kind="uniform", atol=1e-7, rtol=0)
    c = hip.get_counters()
    assert c["capture_wave_timeouts"] == 0 and hip.capture_wave


@pytest.mark.parametrize("n", [1, 9, 130])
@pytest.mark.parametrize("name", ["objlock_train", "combined_big_reach"])
def test_capture_wave_with_in_kernel_resets_and_ragged_env_counts(oracle, name, n, monkeypatch):
    """The two-wave kernels where the hand-off of pre-simulated episode starts is off (FWSIM_NO_SHADOW=1): every auto-reset then
    runs its warm-up inside the step wave, whose captures go through the mailbox and are waited for at once (the warm-up's last
    sub-step reads its frame) -- and at env counts that leave the last workgroup partly or almost wholly empty."""
    cfg, kind = CASES[name]
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8"); monkeypatch.setenv("FWSIM_CAPTURE_WAVE", "1"); monkeypatch.setenv("FWSIM_NO_SHADOW", "1")
    hip = P.FixedwingVecEnv(cfg, n, seed=99)
    assert hip.capture_wave
    ora = oracle.OracleEnv(cfg, n, seed=99)
    obj = cfg.task == K.FW_TASK_OBJLOCK
    worst = run_lockstep(hip, ora, 400, np.random.default_rng(3), kind="uniform", atol=2e-5 if obj else 1e-7, rtol=0, state_atol=1e-7)
    c = hip.get_counters()
    assert c["capture_wave_timeouts"] == 0 and c["shadow_hits"] == 0
    if n >= 9:
        assert c["resets"] > 0 and c["fallbacks"] == c["resets"], c      # the case was meant to reset, and in the kernel


@pytest.mark.parametrize("task,n,steps", [("objlock", 4096, 10), ("combined", 2048, 10), ("combined", 16384, 8)])
def test_baseline_size_camera_tasks_against_oracle(oracle, task, n, steps):
    """configs[2] of BASELINE.json (ObjLock, 4096 envs: train/train_objlock.py:27-86), configs[4]'s per-GPU share (combined env,
    2048 envs: train/train_Fixedwing_Waypoints_ObjLock.py:35-92) and its whole 16 384 envs on the 8-lane mapping fw_create keeps
    them on -- in lockstep with the oracle over >= 8 agent steps: the camera fires every 3rd agent step (capture interval 12
    Aviary steps, 4 per agent step), so the window spans capture sub-steps and stale-frame sub-steps alike."""
    cfg = K.train_objlock_config() if task == "objlock" else K.train_waypoint_objlock_config()
    hip = P.FixedwingVecEnv(cfg, n, seed=42)
    assert hip.lanes_per_env == 8
    ora = oracle.OracleEnv(cfg, n, seed=42)
    obj = task == "objlock"
    worst = run_lockstep(hip, ora, steps, np.random.default_rng(0), kind="gentle", atol=2e-5 if obj else 1e-7, rtol=0, state_atol=1e-7)
    assert worst["obs"] < (2e-5 if obj else 1e-7) and worst["state"] < 1e-7
    c = hip.get_counters()
    assert c["launches"] >= steps


def test_gimbal_guard_branch(oracle, lanes):
    """pybullet's getEulerFromQuaternion switches formulas at |sin(pitch)| >= 0.99999; the
    kernel takes the Euler->quaternion round trip only there.  Force it."""
    cfg = K.waypoints_config(sparse_reward=False, num_targets=3, angle_representation="euler", motor_noise=False,
                             flight_dome_size=1e4)
    n = 64
    hip = P.FixedwingVecEnv(cfg, n, seed=9); ora = oracle.OracleEnv(cfg, n, seed=9)
    hip.reset_tensor(); ora.reset()
    s = ora.get_state()
    rng = np.random.default_rng(3)
    for i in range(n):
        off = 10 ** (rng.uniform(-9, -2.6) if (i // 2) % 2 else rng.uniform(-2.2, -1.0))   # inside / outside the guard (0.00447 rad)
        pitch = (np.pi / 2 - off) * (1 if i % 2 else -1)
        e = [rng.uniform(-3, 3), pitch, rng.uniform(-3, 3)]
        s[i, K.S_QUAT:K.S_QUAT + 4] = oracle.quat_from_euler(e)
        s[i, K.S_POS + 2] = 500.0
    hip.set_state(s); ora.set_state(s)
    oh, oo = hip.observe_tensor().cpu().numpy(), ora.observe()
    guarded = np.abs(oo[:, 4]) == 0.5 * np.pi
    assert 5 < guarded.sum() < n - 5, "test must cover both sides of the guard"
    np.testing.assert_allclose(oh, oo, rtol=0, atol=1e-9)
    import torch
    a = seeded_actions(rng, n, "gentle")
    o2 = ora.step(a)
    hip.step_tensor(torch.as_tensor(a, device=hip.device))
    np.testing.assert_allclose(hip.obs.cpu().numpy(), o2[0], rtol=0, atol=1e-8)


def test_truncation_timing_matches_reference_quirk(lanes):
    """strict '>' and a post-incremented counter => first truncation on call max_steps+2
    (envs/fixedwing_envs/fixedwing_base_env.py:299,346)."""
    import torch
    cfg = K.waypoints_config(max_duration_seconds=1.0, flight_dome_size=1e6, num_targets=1, goal_reach_distance=1e-9,
                             angle_representation="euler", motor_noise=False, auto_reset=False)
    env = P.FixedwingVecEnv(cfg, 3, seed=0)
    env.reset_tensor()
    s = env.get_state(); s[:, K.S_POS + 2] = 500.0; env.set_state(s)
    a = torch.zeros((3, 4), dtype=torch.float64, device=env.device)
    first = None
    for call in range(1, 40):
        env.step_tensor(a)
        assert not env.terminated.any().item()
        if env.truncated.all().item():
            first = call
            break
    assert first == 32


def test_lane_mappings_agree_with_each_other(monkeypatch):
    """The two lane mappings sum the surface wrench in a different order; they must still agree to rounding."""
    import torch
    cfg = K.train_waypoints_v3_config()
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "1"); a = P.FixedwingVecEnv(cfg, 1000, seed=5)
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8"); b = P.FixedwingVecEnv(cfg, 1000, seed=5)
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    g = torch.Generator(device="cpu").manual_seed(3)
    for _ in range(100):
        act = (torch.rand((1000, 4), generator=g, dtype=torch.float64) * 2 - 1).to(a.device)
        a.step_tensor(act); b.step_tensor(act)
        assert torch.equal(a.terminated, b.terminated) and torch.equal(a.truncated, b.truncated)
        assert torch.equal(a.info, b.info)
        assert (a.obs - b.obs).abs().max().item() < 1e-9 and torch.equal(a.rewards, b.rewards)


@pytest.mark.parametrize("which", ["waypoints_gust_force", "objlock", "combined"])
def test_shadow_warmup_equals_inline_warmup(monkeypatch, lanes, which):
    """The background (shadow) warm-up of the next episode and the in-kernel fallback are the same
    computation (the compiler may contract FMAs differently at the two inlining sites, so they agree to
    rounding, not bit for bit): identical episode boundaries, trajectories within 1e-9 across hundreds of resets."""
    import torch
    cfg = {"waypoints_gust_force": K.waypoints_config(sparse_reward=False, num_targets=4, goal_reach_distance=12.0,
                                                      angle_representation="euler", wind_config=GUST_FORCE),
           "objlock": K.train_objlock_config(flight_dome_size=120.0, duck_camera_capture_interval_steps=3),
           "combined": K.train_waypoint_objlock_config(goal_reach_distance=20.0)}[which]
    n = 520
    a = P.FixedwingVecEnv(cfg, n, seed=6)
    monkeypatch.setenv("FWSIM_NO_SHADOW", "1")
    b = P.FixedwingVecEnv(cfg, n, seed=6)
    monkeypatch.delenv("FWSIM_NO_SHADOW")
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    g = torch.Generator(device="cpu").manual_seed(11)
    ends = 0
    for t in range(260):
        act = (torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).to(a.device)
        a.step_tensor(act); b.step_tensor(act)
        assert torch.equal(a.terminated, b.terminated) and torch.equal(a.truncated, b.truncated) and torch.equal(a.info, b.info), t
        tol = 2e-5 if cfg.task == K.FW_TASK_OBJLOCK else 1e-9          # ObjLock obs are float32-rounded
        assert (a.obs - b.obs).abs().max().item() < tol and (a.rewards - b.rewards).abs().max().item() < 1e-9, t
        ends += int((a.terminated | a.truncated).sum())
    assert ends > 300
    np.testing.assert_allclose(a.get_state(), b.get_state(), rtol=0, atol=1e-9)
    # an explicit reset / re-seed in the middle invalidates the shadows consistently
    a.seed(99); b.seed(99)
    assert torch.equal(a.reset_tensor(), b.reset_tensor())
    for t in range(40):
        act = (torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).to(a.device)
        a.step_tensor(act); b.step_tensor(act)
        assert (a.obs - b.obs).abs().max().item() < 2e-5, t


def test_sharding_is_world_size_independent(lanes):
    """rank r of a sharded job (global_env_offset = r*N_local) reproduces envs [r*N_local, ...) of one big job, bit for bit."""
    import torch
    cfg = K.train_waypoints_v3_config()
    big = P.FixedwingVecEnv(cfg, 4096, seed=42)
    shard = P.FixedwingVecEnv(cfg, 1024, seed=42, global_env_offset=3072)
    ob, os_ = big.reset_tensor().clone(), shard.reset_tensor().clone()
    assert torch.equal(ob[3072:], os_)
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(40):
        a = (torch.rand((4096, 4), generator=g, dtype=torch.float64) * 2 - 1).to(big.device)
        big.step_tensor(a); shard.step_tensor(a[3072:].contiguous())
        assert torch.equal(big.obs[3072:], shard.obs) and torch.equal(big.rewards[3072:], shard.rewards)
        assert torch.equal(big.terminated[3072:], shard.terminated) and torch.equal(big.info[3072:], shard.info)


@pytest.mark.parametrize("task", ["objlock", "combined"])
def test_camera_frame_does_not_depend_on_the_wave_neighbours(task, monkeypatch):
    """8-lane mapping: the camera is run by the whole wave -- the envs due at a sub-step share the 64 lanes (lane sets), the
    cylinder slices and the row chunks go to whichever 8-lane group is free.  What an env sees must not depend on that:
    an env alone in its wave (all 64 lanes work for it) and the same env among 63 others produce the same bits, through
    captures, strikes / collisions and auto-resets (the pre-simulated episodes are built by the same code)."""
    import torch
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "8")
    if task == "objlock":
        cfg = K.train_objlock_config(num_obstacles=14, obstacle_radius=2.0, duck_camera_capture_interval_steps=1)
    else:
        cfg = K.train_waypoint_objlock_config(duck_camera_capture_interval_steps=1)
    n, seed, picks = 64, 5, (0, 5, 17, 42, 63)
    big = P.FixedwingVecEnv(cfg, n, seed=seed)
    ones = {k: P.FixedwingVecEnv(cfg, 1, seed=seed, global_env_offset=k) for k in picks}
    ob = big.reset_tensor().clone()
    for k, e in ones.items():
        assert torch.equal(e.reset_tensor()[0], ob[k]), k
    g = torch.Generator(device="cpu").manual_seed(3)
    ends = 0
    for t in range(150):
        a = (torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1)
        a[:, 3] = a[:, 3].abs()                                      # keep them flying: more frames with something in view
        a = a.to(big.device)
        big.step_tensor(a)
        for k, e in ones.items():
            e.step_tensor(a[k:k + 1].contiguous())
            assert torch.equal(e.obs[0], big.obs[k]) and torch.equal(e.rewards[0], big.rewards[k]), (t, k)
            assert torch.equal(e.terminated[0], big.terminated[k]) and torch.equal(e.info[0], big.info[k]), (t, k)
        ends += int((big.terminated | big.truncated).bool()[list(picks)].sum())
    sb = big.get_state()
    for k, e in ones.items():
        assert np.array_equal(e.get_state()[0], sb[k]), k
    c = big.get_counters()
    assert c["resets"] > 0 and c["shadow_hits"] > 0


def test_properties_at_baseline_size(lanes):
    """Size-independent invariants at N=4096 over 300 steps of random actions."""
    import torch
    cfg = K.train_waypoints_v3_config()
    env = P.FixedwingVecEnv(cfg, 4096, seed=7)
    twin = P.FixedwingVecEnv(cfg, 4096, seed=7)
    env.reset_tensor(); twin.reset_tensor()
    g = torch.Generator(device="cpu").manual_seed(1)
    ends = 0
    for t in range(300):
        a = (torch.rand((4096, 4), generator=g, dtype=torch.float64) * 2 - 1).to(env.device)
        obs, rew, term, trunc = env.step_tensor(a)
        twin.step_tensor(a)
        assert torch.equal(obs, twin.obs) and torch.equal(rew, twin.rewards)              # deterministic
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        done = (term | trunc).bool()
        ends += int(done.sum())
        # a done env was auto-reset: its new obs carries a zero action and step_count restarts
        assert (obs[done][:, 12:16] == 0).all()
        assert torch.equal(obs[~done][:, 12:16], a[~done])
        # terminated <=> collision or out-of-bounds; sparse reward values are {-0.1, 100, -100}
        info = env.info
        assert torch.equal(term.bool(), (info[:, K.INFO_COLLISION] | info[:, K.INFO_OUT_OF_BOUNDS]).bool())
        assert bool(((rew == -0.1) | (rew == 100.0) | (rew == -100.0)).all())
        assert bool((rew[term.bool()] != -0.1).all())
    assert ends > 1000
    s = env.get_state()
    np.testing.assert_allclose(np.linalg.norm(s[:, K.S_QUAT:K.S_QUAT + 4], axis=1), 1.0, atol=1e-14)
    assert np.all(np.linalg.norm(s[:, K.S_POS:K.S_POS + 3], axis=1) <= 100.0 + 1e-9)       # inside the dome
    assert np.all((s[:, K.S_FLAGS].astype(int) & 0xFF) == 0) and np.all(s[:, K.S_STEP_COUNT] <= 300)
    # observation of position/actuators is the state itself
    np.testing.assert_array_equal(env.obs.cpu().numpy()[:, 9:12], s[:, K.S_POS:K.S_POS + 3])
    np.testing.assert_array_equal(env.obs.cpu().numpy()[:, 16:22], s[:, K.S_ACT:K.S_ACT + 6])


def test_reset_mask_and_seed(oracle, lanes):
    import torch
    cfg = K.train_waypoints_v3_config(motor_noise=False)
    n = 130
    env = P.FixedwingVecEnv(cfg, n, seed=11); ora = oracle.OracleEnv(cfg, n, seed=11)
    env.reset_tensor(); ora.reset()
    rng = np.random.default_rng(2)
    for _ in range(5):
        a = seeded_actions(rng, n, "gentle")
        env.step_tensor(torch.as_tensor(a, device=env.device)); ora.step(a)
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    before = env.get_state()
    oh = env.reset_tensor(torch.as_tensor(mask)).cpu().numpy()
    oo = ora.reset(mask)
    after = env.get_state()
    keep = mask == 0
    np.testing.assert_array_equal(after[keep], before[keep])
    assert np.all(after[~keep, K.S_EPISODE] == before[~keep, K.S_EPISODE] + 1)
    np.testing.assert_allclose(oh, oo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(after, ora.get_state(), rtol=0, atol=1e-9)
    # re-seeding restarts the scenario stream: same seed -> same first episode
    env.seed(99); ora.seed(99)
    a1 = env.reset_tensor().clone()
    env.seed(99)
    assert torch.equal(a1, env.reset_tensor())
    np.testing.assert_allclose(a1.cpu().numpy(), ora.reset(), rtol=0, atol=1e-9)


def test_abi_writes_stay_inside_the_buffers(lanes):
    """Call fw_step directly with sentinel-padded buffers for N not a multiple of the wave size."""
    import torch
    cfg = K.train_waypoints_v3_config()
    n, d, pad = 65, 28, 64
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.fw_create(C.byref(cfg), n, torch.cuda.current_device(), 5, 0, C.byref(h)))
    dev = torch.device("cuda")
    obs = torch.full((n * d + pad,), -7.0, dtype=torch.float64, device=dev)
    tobs = torch.full((n * d + pad,), -7.0, dtype=torch.float64, device=dev)
    rew = torch.full((n + pad,), -7.0, dtype=torch.float64, device=dev)
    term = torch.full((n + pad,), 77, dtype=torch.uint8, device=dev)
    trunc = torch.full((n + pad,), 77, dtype=torch.uint8, device=dev)
    info = torch.full(((n) * K.FW_INFO_DIM + pad,), -7, dtype=torch.int32, device=dev)
    act = torch.zeros((n, 4), dtype=torch.float64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.fw_reset(h, None, None, p(obs), None), h)
    for _ in range(3):
        _lib.check(L.fw_step(h, p(act), p(obs), p(rew), p(term), p(trunc), p(tobs), p(info), None), h)
    torch.cuda.synchronize()
    assert (obs[n * d:] == -7.0).all() and (tobs[n * d:] == -7.0).all() and (rew[n:] == -7.0).all()
    assert (term[n:] == 77).all() and (trunc[n:] == 77).all() and (info[n * K.FW_INFO_DIM:] == -7).all()
    assert (obs[:n * d] != -7.0).any() and (rew[:n] == -0.1).all()
    # NULL optional outputs are accepted; NULL mandatory ones are FW_EINVAL, not a crash
    assert L.fw_step(h, p(act), p(obs), p(rew), p(term), p(trunc), None, None, None) == K.FW_OK
    assert L.fw_step(h, None, p(obs), p(rew), p(term), p(trunc), None, None, None) == K.FW_EINVAL
    assert b"non-NULL" in L.fw_last_error(h)
    torch.cuda.synchronize()
    assert L.fw_destroy(h) == K.FW_OK


def test_vecenv_numpy_surface():
    """SB3 VecEnv duck type: shapes, dones, terminal_observation, TimeLimit.truncated."""
    env = P.FixedwingWaypointsVecEnv(32, sparse_reward=True, num_targets=8, goal_reach_distance=4.0,
                                     angle_representation="euler", max_duration_seconds=1.0, flight_dome_size=1e4,
                                     seed=42)
    assert env.num_envs == 32 and env.observation_space.shape == (28,) and env.action_space.shape == (4,)
    assert env.observation_space.dtype == np.float64 and float(env.action_space.low.min()) == -1.0
    obs = env.reset()
    assert obs.shape == (32, 28) and obs.dtype == np.float64
    saw_trunc = False
    for _ in range(40):
        act = np.stack([env.action_space.sample() for _ in range(32)]) * 0.1
        obs, rew, dones, infos = env.step(act)
        assert obs.shape == (32, 28) and rew.shape == (32,) and dones.dtype == bool and len(infos) == 32
        for i, info in enumerate(infos):
            assert {"out_of_bounds", "collision", "env_complete", "num_targets_reached", "TimeLimit.truncated"} <= set(info)
            if dones[i]:
                assert info["terminal_observation"].shape == (28,)
                if info["TimeLimit.truncated"]:
                    saw_trunc = True
                    assert not (info["collision"] or info["out_of_bounds"])
            else:
                assert "terminal_observation" not in info
    assert saw_trunc
    assert env.env_is_wrapped(object) == [False] * 32 and env.get_attr("num_envs", [0, 1]) == [32, 32]
    env.close()


def test_f32_throughput_mode_single_step_error(oracle):
    """fp32 kernel vs the fp64 oracle from identical states: one agent step stays within 2e-3."""
    import torch
    cfg64 = K.train_waypoints_v3_config(motor_noise=False)
    cfg32 = K.train_waypoints_v3_config(motor_noise=False, dtype="float32")
    n = 512
    ora = oracle.OracleEnv(cfg64, n, seed=3); ora.reset()
    rng = np.random.default_rng(4)
    for _ in range(20):
        ora.step(seeded_actions(rng, n, "gentle"))
    hip = P.FixedwingVecEnv(cfg32, n, seed=3); hip.reset_tensor()
    s = ora.get_state(); hip.set_state(s)
    s32 = hip.get_state(); ora.set_state(s32)          # start both from the float32-rounded state
    a = seeded_actions(rng, n, "gentle").astype(np.float32)
    oo, ro, te, tr, _, _ = ora.step(a.astype(np.float64))
    hip.step_tensor(torch.as_tensor(a, device=hip.device))
    same = (hip.terminated.cpu().numpy() == te) & (hip.truncated.cpu().numpy() == tr) & ~(te | tr).astype(bool)
    assert same.mean() > 0.97
    err = np.abs(hip.obs.cpu().numpy().astype(np.float64) - oo)[same]
    assert err.max() < 2e-3, err.max()
    q = hip.get_state()[:, K.S_QUAT:K.S_QUAT + 4]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("task", ["objlock", "combined"])
def test_f32_throughput_mode_camera_tasks_single_step_error(oracle, task, lanes):
    """The float instantiation of the camera kernels (the wave-level camera keeps its row buffer and its nearest fragments as
    ordered 32-bit patterns there, and its exact sums in ds_add_f32): one agent step from the oracle's state, cylinders and
    ducks in view, stays close to the fp64 oracle; the frame's pixel statistics (centroid, area: sums of integers) agree to
    float rounding wherever both saw the duck."""
    import torch
    kw = dict(motor_noise=False, duck_camera_capture_interval_steps=1, wind_config=None)
    mk = (lambda **o: K.train_objlock_config(num_obstacles=12, **kw, **o)) if task == "objlock" else (lambda **o: K.train_waypoint_objlock_config(**kw, **o))
    n = 256
    ora = oracle.OracleEnv(mk(), n, seed=3); ora.reset()
    rng = np.random.default_rng(4)
    for _ in range(30):
        ora.step(seeded_actions(rng, n, "gentle"))
    hip = P.FixedwingVecEnv(mk(dtype="float32"), n, seed=3); hip.reset_tensor()
    hip.set_state(ora.get_state())
    ora.set_state(hip.get_state())                       # start both from the float32-rounded state
    a = seeded_actions(rng, n, "gentle").astype(np.float32)
    oo, ro, te, tr, _, _ = ora.step(a.astype(np.float64))
    hip.step_tensor(torch.as_tensor(a, device=hip.device))
    same = (hip.terminated.cpu().numpy() == te) & (hip.truncated.cpu().numpy() == tr) & ~(te | tr).astype(bool)
    assert same.mean() > 0.95
    assert torch.isfinite(hip.obs).all() and torch.isfinite(hip.rewards).all()
    T0 = K.S_TASK
    sh, so = hip.get_state()[same], ora.get_state()[same]
    fh, fo = sh[:, T0 + K.ST_FRAME:T0 + K.ST_FRAME + 8], so[:, T0 + K.ST_FRAME:T0 + K.ST_FRAME + 8]
    both = (fh[:, 0] > 0) & (fo[:, 0] > 0)
    assert (fh[:, 0] == fo[:, 0]).mean() > 0.97              # visibility agrees (a mask of a few pixels may flip in float)
    if both.any():
        assert np.abs(fh[both, 1:4] - fo[both, 1:4]).max() < 2e-2       # centroid / area of the mask
    zones_h, zones_o = fh[:, 5:8], fo[:, 5:8]
    ok = (zones_h > 0) & (zones_o > 0)
    assert ok.any()
    rel = np.abs(zones_h[ok] - zones_o[ok]) / zones_o[ok]
    assert np.median(rel) < 1e-2, np.median(rel)                       # zone depths in metres (float depth buffer near the far plane is coarse)


@pytest.mark.parametrize("n", [1, 9, 65, 130])
@pytest.mark.parametrize("task", ["waypoints_wind", "objlock"])
def test_ragged_env_counts_match_the_oracle(oracle, lanes, n, task):
    """Env counts that do not fill a tile / a wave / the 64-padding (partly and wholly inactive tiles, single env):
    lockstep with the oracle incl. auto-resets, the background warm-up hand-off and get/set_state through the tiled layout."""
    import pyflyt_drone_amd as P
    from pyflyt_drone_amd import config as K
    if task == "objlock":
        cfg = K.train_objlock_config(max_duration_seconds=2.0)
    else:
        cfg = K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND, flight_dome_size=25.0, max_duration_seconds=3.0)
    hip, ora = P.FixedwingVecEnv(cfg, n, seed=11), oracle.OracleEnv(cfg, n, seed=11)
    w = run_lockstep(hip, ora, 80, np.random.default_rng(n), atol=1e-7, state_atol=2e-5 if task == "objlock" else 1e-7)
    assert w["dones"] >= 1
    st = hip.get_state()
    hip.set_state(st)
    np.testing.assert_array_equal(hip.get_state(), st)


@pytest.mark.parametrize("task", ["waypoints_gust", "objlock", "combined"])
def test_caller_supplied_scenario_replaces_the_env_draw(oracle, lanes, task):
    """fw_reset(handle, mask, scenario, obs) of SURVEY section 8(b): targets / duck / obstacles / wind handed in by the caller
    (e.g. what a PyFlyt run sampled) start the episode instead of the env's own draw -- in the kernel and in the oracle
    alike: the state shows them, the warm-up and the first observation are computed under them, the lockstep trace that
    follows agrees, and later auto-resets go back to the env's own streams."""
    import torch
    rng = np.random.default_rng(21)
    n = 130
    if task == "waypoints_gust":
        cfg = K.waypoints_config(sparse_reward=False, num_targets=5, goal_reach_distance=10.0, angle_representation="euler",
                                 wind_config=GUST_FORCE, max_duration_seconds=4.0)
    elif task == "objlock":
        cfg = K.objlock_config(flight_dome_size=150.0, max_duration_seconds=4.0, num_obstacles=6, obstacle_safe_distance_m=40.0,
                               duck_camera_capture_interval_steps=2, angle_representation="euler", camera_resolution=128,
                               wind_config=GUST_FORCE)
    else:
        cfg = K.train_waypoint_objlock_config(goal_reach_distance=25.0, max_duration_seconds=4.0)
    nt = cfg.num_targets
    sc = dict(wind_base=rng.uniform(-4, 4, (n, 3)) * [1, 1, 0.05], gust_amp=rng.uniform(0, 2, (n, 3)) * [1, 1, 0.1],
              gust_phase=rng.uniform(0, 2 * np.pi, n))
    if nt:
        sc["targets"] = np.concatenate([rng.uniform(-60, 60, (n, nt, 2)), rng.uniform(5, 40, (n, nt, 1))], axis=2)
    if cfg.task != K.FW_TASK_WAYPOINTS:
        nob = rng.integers(0, 7, n).astype(np.int32)
        ob = np.concatenate([rng.uniform(-70, 70, (n, 20, 2)), rng.uniform(10, 30, (n, 20, 1))], axis=2)
        ob[np.linalg.norm(ob[:, :, :2], axis=2) < 15] += 40.0            # keep the start area free
        sc["obstacles"], sc["num_obstacles"] = ob, nob
        if cfg.task == K.FW_TASK_OBJLOCK:
            sc["duck_pos"] = np.concatenate([rng.uniform(-70, 70, (n, 2)), np.full((n, 1), 0.05)], axis=1)
    hip, ora = P.FixedwingVecEnv(cfg, n, seed=31), oracle.OracleEnv(cfg, n, seed=31)
    osc, keep = K.make_scenario(n, **sc)
    oh = hip.reset_tensor(scenario=sc).cpu().numpy()
    oo = ora.reset(scenario=osc)
    np.testing.assert_allclose(oh, oo, rtol=0, atol=2e-5 if cfg.task == K.FW_TASK_OBJLOCK else 1e-9)
    st = hip.get_state()
    np.testing.assert_allclose(st, ora.get_state(), rtol=0, atol=1e-9)
    np.testing.assert_array_equal(st[:, K.S_WIND:K.S_WIND + 3], sc["wind_base"])
    np.testing.assert_array_equal(st[:, K.S_WIND + 6], sc["gust_phase"])
    if nt:
        np.testing.assert_array_equal(st[:, K.S_TARGETS:K.S_TARGETS + 3 * nt].reshape(n, nt, 3), sc["targets"])
    if cfg.task != K.FW_TASK_WAYPOINTS:
        T0 = K.S_TASK
        np.testing.assert_array_equal(st[:, T0 + K.ST_NUM_OBST], nob)
        for i in range(n):
            np.testing.assert_array_equal(st[i, T0 + K.ST_OBST:T0 + K.ST_OBST + 3 * nob[i]].reshape(-1, 3), ob[i, :nob[i]])
        if cfg.task == K.FW_TASK_OBJLOCK:
            np.testing.assert_array_equal(st[:, T0:T0 + 3], sc["duck_pos"])
        else:                                                              # the duck sits under the supplied last waypoint
            np.testing.assert_array_equal(st[:, T0:T0 + 2], sc["targets"][:, nt - 1, :2])
    # the trace that follows, through the first auto-resets (which draw their own scenarios again)
    dones = 0
    for t in range(150):
        a = seeded_actions(rng, n, "gentle")
        o_obs, o_rew, o_term, o_trunc, o_tobs, o_info = ora.step(a)
        hip.step_tensor(torch.as_tensor(a, device=hip.device))
        assert np.array_equal(hip.terminated.cpu().numpy(), o_term) and np.array_equal(hip.truncated.cpu().numpy(), o_trunc), t
        assert np.array_equal(hip.info.cpu().numpy(), o_info), t
        np.testing.assert_allclose(hip.obs.cpu().numpy(), o_obs, rtol=0, atol=2e-5 if cfg.task == K.FW_TASK_OBJLOCK else 1e-7, err_msg=f"obs {t}")
        np.testing.assert_allclose(hip.rewards.cpu().numpy(), o_rew, rtol=0, atol=1e-7, err_msg=f"reward {t}")
        dones += int((o_term | o_trunc).sum())
    assert dones >= n, "every env was meant to end its supplied episode and auto-reset"
    # a masked reset with a scenario only touches the masked envs
    mask = (rng.uniform(size=n) < 0.4).astype(np.uint8)
    before = hip.get_state()
    hip.reset_tensor(torch.as_tensor(mask), scenario=sc); ora.reset(mask, scenario=osc)
    after = hip.get_state()
    np.testing.assert_array_equal(after[mask == 0], before[mask == 0])
    np.testing.assert_allclose(after, ora.get_state(), rtol=0, atol=1e-9)
    del keep


def test_camera_frame_statistics_match_the_oracle_on_directed_poses(oracle, lanes):
    """The image functionals of envs/fixedwing_objlock_env.py:662-743 (mask mean / pixel count / min depth-buffer value,
    zone means of the depth BUFFER over the non-duck pixels of row h//2): the kernel's closed-form row intervals against the
    oracle's literal per-pixel loops, on poses chosen to hit the hard cases -- rolled / pitched cameras, ducks clipped by
    the image border, ducks crossing the middle row, ducks straddling the far plane (255 m), cylinders inside the zones,
    in front of the duck and behind the camera."""
    import math
    import torch
    T0 = K.S_TASK
    rng = np.random.default_rng(77)
    for res, nobs in ((480, 20), (128, 5), (65, 0), (1024, 20)):        # 1024 columns: the camera's LDS map is 89 KB (opt-in above 64 KB)
        cfg = K.objlock_config(agent_hz=120, motor_noise=False, auto_reset=False, angle_representation="euler",
                               duck_camera_capture_interval_steps=1, flight_dome_size=1e5, num_obstacles=max(nobs, 1),
                               obstacle_radius=2.0, duck_global_scaling=60.0, camera_resolution=res)
        n = 256
        hip, ora = P.FixedwingVecEnv(cfg, n, seed=5), oracle.OracleEnv(cfg, n, seed=5)
        hip.reset_tensor(); ora.reset()
        s = ora.get_state()
        for i in range(n):
            kind = i % 4
            dist = rng.uniform(12, 200) if kind != 3 else rng.uniform(249.0, 259.0)        # kind 3: around the far plane
            yaw, roll, pitch = rng.uniform(-3, 3), rng.uniform(-0.7, 0.7), rng.uniform(-0.3, 0.3)
            pos = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(4, 45)])
            bearing = yaw + rng.uniform(-0.8, 0.8)
            if kind in (1, 3):                                                              # aimed at the duck: mask on / near row h//2
                bearing = yaw + rng.uniform(-0.15, 0.15); roll = rng.uniform(-0.3, 0.3)
                pitch = math.atan2(pos[2], dist) - math.radians(5.0) + rng.uniform(-0.02, 0.02)
            s[i, K.S_POS:K.S_POS + 3] = pos
            s[i, K.S_QUAT:K.S_QUAT + 4] = oracle.quat_from_euler([roll, pitch, yaw])
            s[i, K.S_VEL:K.S_VEL + 3] = oracle.mat_from_quat(s[i, K.S_QUAT:K.S_QUAT + 4]) @ np.array([20.0, 0, 0])
            s[i, K.S_OMEGA:K.S_OMEGA + 3] = 0
            s[i, T0:T0 + 3] = [pos[0] + dist * math.cos(bearing), pos[1] + dist * math.sin(bearing), 0.05]
            k = int(rng.integers(0, nobs + 1))
            s[i, T0 + K.ST_NUM_OBST] = k
            for o in range(k):
                ang, d = yaw + rng.uniform(-math.pi, math.pi) * (1.0 if o % 3 == 0 else 0.25), rng.uniform(8, 120)
                s[i, T0 + K.ST_OBST + 3 * o:T0 + K.ST_OBST + 3 * o + 3] = [pos[0] + d * math.cos(ang), pos[1] + d * math.sin(ang), rng.uniform(10, 60)]
        hip.set_state(s); ora.set_state(s)
        a = np.zeros((n, 4))
        ora.step(a); hip.step_tensor(torch.as_tensor(a, device=hip.device))                 # 2 ticks, then a capture
        fh = hip.get_state()[:, T0 + K.ST_FRAME:T0 + K.ST_FRAME + 8]
        fo = ora.get_state()[:, T0 + K.ST_FRAME:T0 + K.ST_FRAME + 8]
        assert np.array_equal(fh[:, 0], fo[:, 0]), np.nonzero(fh[:, 0] != fo[:, 0])[0]
        np.testing.assert_allclose(fh[:, 1:4], fo[:, 1:4], rtol=0, atol=1e-15, err_msg=f"mask statistics, res {res}")
        np.testing.assert_allclose(fh[:, 4:], fo[:, 4:], rtol=1e-9, atol=1e-9, err_msg=f"depths, res {res}")
        vis = fo[:, 0] > 0
        assert vis.sum() > n // 4 and (~vis).sum() > 10
        if nobs:
            assert (np.abs(fo[:, 5:8] - fo[:, 5:8].mean(1, keepdims=True)).max(1) > 1.0).sum() > 20, "cylinders were meant to change zone depths"
        far_cases = fo[3::4]
        assert 0 < (far_cases[:, 0] > 0).sum() < len(far_cases), "far-plane cases must include visible and clipped ducks"


def test_fw_create_picks_the_two_wave_build_between_8k_and_16k_envs(oracle, monkeypatch):
    """N = 16 384 waypoint envs (configs[4]'s one-GPU size) on the 8-lane mapping is 2048 waves; the
    full-register-file build holds one wave per SIMD (two rounds: the 8 k -> 16 k cliff), so fw_create switches to the build capped
    at 256 registers there.  The choice is visible (lanes_per_env stays 8) and the mapping it picks follows the oracle."""
    monkeypatch.delenv("FWSIM_LANES_PER_ENV", raising=False); monkeypatch.delenv("FWSIM_G8_WAVES", raising=False)
    cfg = K.train_waypoints_v3_config()
    n = 16384
    hip, ora = P.FixedwingVecEnv(cfg, n, seed=42), oracle.OracleEnv(cfg, n, seed=42)
    assert hip.lanes_per_env == 8 and hip.g8_waves == 2
    run_lockstep(hip, ora, 6, np.random.default_rng(0), kind="uniform", atol=1e-7, rtol=0)
