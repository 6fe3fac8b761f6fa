"""fw_render (the FPV image a CNN front end consumes) against the oracle's literal per-pixel render, through the C ABI.
Reference: what Camera.capture_image() hands the env (envs/fixedwing_objlock_env.py:603-622) and what the CNN path feeds a
network (envs/fixedwing_envs/objlock_yolo_env.py:646-716).  Duck mask bit-exact; depth-buffer channel within 1e-7 (one float32
ulp of a value in [0, 1] is 6e-8: both sides round the same double expression to float32)."""
import math

import numpy as np
import pytest
import torch

import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K

pytestmark = pytest.mark.gpu
T0 = K.S_TASK


@pytest.fixture(params=[1, 8], ids=["lane_per_env", "8_lanes_per_env"])
def lanes(request, monkeypatch):
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", str(request.param))     # the render reads the state through the handle's tiling
    return request.param


def _directed_state(oracle, s, rng, nobs):
    n = s.shape[0]
    for i in range(n):
        kind = i % 4
        dist = rng.uniform(12, 200) if kind != 3 else rng.uniform(249.0, 259.0)        # kind 3: around the far plane
        yaw, roll, pitch = rng.uniform(-3, 3), rng.uniform(-0.7, 0.7), rng.uniform(-0.3, 0.3)
        pos = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(4, 45)])
        bearing = yaw + rng.uniform(-0.8, 0.8)
        if kind in (1, 3):                                                              # aimed at the duck
            bearing = yaw + rng.uniform(-0.15, 0.15); roll = rng.uniform(-0.3, 0.3)
            pitch = math.atan2(pos[2], dist) - math.radians(5.0) + rng.uniform(-0.02, 0.02)
        s[i, K.S_POS:K.S_POS + 3] = pos
        s[i, K.S_QUAT:K.S_QUAT + 4] = oracle.quat_from_euler([roll, pitch, yaw])
        s[i, T0:T0 + 3] = [pos[0] + dist * math.cos(bearing), pos[1] + dist * math.sin(bearing), 0.05]
        k = int(rng.integers(0, nobs + 1))
        s[i, T0 + K.ST_NUM_OBST] = k
        for o in range(k):
            ang, d = yaw + rng.uniform(-math.pi, math.pi) * (1.0 if o % 3 == 0 else 0.25), rng.uniform(8, 120)
            s[i, T0 + K.ST_OBST + 3 * o:T0 + K.ST_OBST + 3 * o + 3] = [pos[0] + d * math.cos(ang), pos[1] + d * math.sin(ang), rng.uniform(10, 60)]
    return s


@pytest.mark.parametrize("task", ["objlock", "combined"])
def test_render_matches_the_oracle_pixel_by_pixel_on_directed_poses(oracle, lanes, task):
    rng = np.random.default_rng(31)
    for res, nobs in ((64, 20), (33, 5), (128, 0), (7, 20)):        # (7: one ragged tile; 33: tiles of 16, 16 and 1 columns)
        if task == "objlock":
            cfg = K.objlock_config(motor_noise=False, auto_reset=False, flight_dome_size=1e5, num_obstacles=max(nobs, 1), obstacle_radius=2.0,
                                   duck_global_scaling=60.0)
        else:
            cfg = K.train_waypoint_objlock_config(motor_noise=False)
            cfg.auto_reset = 0; cfg.num_obstacles = max(nobs, 1); cfg.flight_dome_size = 1e5
        n = 130                                                     # ragged: 2 full tiles + 2 envs on either mapping
        hip, ora = P.FixedwingVecEnv(cfg, n, seed=5), oracle.OracleEnv(cfg, n, seed=5)
        hip.reset_tensor(); ora.reset()
        s = _directed_state(oracle, ora.get_state(), rng, nobs)
        hip.set_state(s); ora.set_state(s)
        got = hip.render_tensor(res).cpu().numpy()
        want = ora.render(res)
        assert got.shape == (n, 2, res, res) and got.dtype == np.float32
        assert np.array_equal(got[:, 0], want[:, 0]), f"duck mask differs in {(got[:, 0] != want[:, 0]).sum()} pixels (res {res})"
        np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=0, atol=1e-7, err_msg=f"depth channel, res {res}")
        ducks = (want[:, 0].sum(axis=(1, 2)) > 0).sum()
        cyl = ((want[:, 1] < 0.999) & (want[:, 0] == 0)).any(axis=(1, 2)).sum()
        assert res < 16 or (ducks >= n // 5 and (nobs == 0 or cyl >= n // 10)), (res, ducks, cyl)      # (the poses are aimed for images of >= 16 pixels)
        hip.close()


def test_render_through_the_lds_stage_equals_the_direct_stores(oracle, monkeypatch):
    """From 64 x 64 pixels on the image leaves through an LDS stage as whole rows (bands of rows that fit the stage); below, and
    above 128 x 128, every lane stores its own pixels.  Same pixels either way, bit for bit: forced stages and forced direct
    stores at sizes with one band, several bands, a ragged last band and a width that is not a multiple of four pixels (the
    scalar copy-out), and with one, two and four waves per env (FWSIM_RENDER_THREADS, a measurement knob)."""
    rng = np.random.default_rng(77)
    cfg = K.train_waypoint_objlock_config(motor_noise=False)
    cfg.auto_reset = 0; cfg.num_obstacles = 20; cfg.flight_dome_size = 1e5
    n = 70
    hip, ora = P.FixedwingVecEnv(cfg, n, seed=9), oracle.OracleEnv(cfg, n, seed=9)
    hip.reset_tensor(); ora.reset()
    s = _directed_state(oracle, ora.get_state(), rng, 20)
    hip.set_state(s); ora.set_state(s)
    for res, stages in ((32, ("1024", "2048")), (48, ("1024", "2048")), (64, ("1024", "2048", "4096")), (66, ("2048",)), (100, ("2048", "4096")), (128, ("2048",))):
        monkeypatch.setenv("FWSIM_RENDER_STAGE", "0"); monkeypatch.delenv("FWSIM_RENDER_THREADS", raising=False)
        direct = hip.render_tensor(res).cpu().numpy()
        if res in (32, 64):                                              # the direct form against the checker once more, the rest against the direct form
            want = ora.render(res)
            assert np.array_equal(direct[:, 0], want[:, 0])
            np.testing.assert_allclose(direct[:, 1], want[:, 1], rtol=0, atol=1e-7)
        for st in stages:
            for th in ("256", "128", "64"):
                monkeypatch.setenv("FWSIM_RENDER_STAGE", st); monkeypatch.setenv("FWSIM_RENDER_THREADS", th)
                got = hip.render_tensor(res).cpu().numpy()
                assert np.array_equal(got, direct), (res, st, th, int((got != direct).sum()))
        monkeypatch.delenv("FWSIM_RENDER_STAGE"); monkeypatch.delenv("FWSIM_RENDER_THREADS")
        assert np.array_equal(hip.render_tensor(res).cpu().numpy(), direct), res          # (the default choice)
    hip.close()


def test_render_from_a_float32_handle_equals_the_float64_handle_on_the_same_values(oracle, lanes):
    """fw_render_kernel<float, .> reads a float32 state (its own gather instructions) and computes in double like the float64
    build: from a state whose every value is float32-representable both handles must write the same image, bit for bit, on the
    direct path (33 pixels) and through the LDS stage (64 pixels); the float64 one is the one the checker is compared with."""
    rng = np.random.default_rng(12)
    n = 70
    envs = []
    for dtype in ("float64", "float32"):
        cfg = K.objlock_config(dtype=dtype, motor_noise=False, auto_reset=False, flight_dome_size=1e5, num_obstacles=20, obstacle_radius=2.0,
                               duck_global_scaling=60.0)
        e = P.FixedwingVecEnv(cfg, n, seed=3); e.reset_tensor(); envs.append(e)
    cfg64 = K.objlock_config(motor_noise=False, auto_reset=False, flight_dome_size=1e5, num_obstacles=20, obstacle_radius=2.0, duck_global_scaling=60.0)
    ora = oracle.OracleEnv(cfg64, n, seed=3); ora.reset()
    s = _directed_state(oracle, ora.get_state(), rng, 20).astype(np.float32).astype(np.float64)
    ora.set_state(s)
    for e in envs:
        e.set_state(s)
    for res in (33, 64):
        a, b = (e.render_tensor(res).cpu().numpy() for e in envs)
        assert np.array_equal(a, b), (res, int((a != b).sum()))
        want = ora.render(res)
        assert np.array_equal(a[:, 0], want[:, 0]) and a[:, 0].sum() > 0
        np.testing.assert_allclose(a[:, 1], want[:, 1], rtol=0, atol=1e-7)
    for e in envs:
        e.close()


def test_render_follows_the_env_through_steps_and_resets(oracle):
    """The image is the scene of the env's CURRENT state: after 40 random agent steps (episodes end, auto-resets place new ducks
    and cylinders) the render still equals the oracle's, which was stepped alongside (its state, not a copy of the kernel's)."""
    cfg = K.train_waypoint_objlock_config(motor_noise=False)
    n = 96
    hip, ora = P.FixedwingVecEnv(cfg, n, seed=9), oracle.OracleEnv(cfg, n, seed=9)
    hip.reset_tensor(); ora.reset()
    rng = np.random.default_rng(2)
    ends = 0
    for t in range(40):
        a = rng.uniform(-1, 1, size=(n, 4)); a[:, 1] -= 0.6 * (t % 3 == 0)          # some dives: episodes end
        a = np.clip(a, -1, 1)
        hip.step_tensor(torch.as_tensor(a, device=hip.device))
        _, _, te, tr, _, _ = ora.step(a)
        ends += int((te | tr).sum())
        if t % 8 == 7:
            got, want = hip.render_tensor(48).cpu().numpy(), ora.render(48)
            same = (got[:, 0] == want[:, 0]).all(axis=(1, 2))
            # poses agree to ~1e-12 after dozens of ticks, not to the bit: a silhouette pixel may flip; almost none do
            assert same.mean() >= 0.97, (t, same.mean())
            np.testing.assert_allclose(got[same, 1], want[same, 1], rtol=0, atol=2e-6)
    assert ends >= 1


def test_render_is_refused_without_a_camera_and_checks_its_buffer():
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 8, seed=0)
    env.reset_tensor()
    with pytest.raises(RuntimeError, match="no camera"):
        env.render_tensor(16)
    cam = P.FixedwingVecEnv(K.train_objlock_config(), 8, seed=0)
    cam.reset_tensor()
    with pytest.raises(ValueError):
        cam.render_tensor(16, out=torch.zeros((8, 2, 16, 15), device=cam.device))
    assert cam.render_tensor(16).shape == (8, 2, 16, 16)


def test_ppo_with_cnn_detector_head_trains_on_the_device_render():
    """configs[4]'s policy form on one GPU: combined waypoint -> duck envs, FPV render every agent step (fw_render), conv
    extractor + MLP on torch-ROCm; the images the policy saw are in the rollout buffer, the update moves the conv weights,
    and the deterministic evaluation loop feeds the policy the eval env's own renders."""
    from pyflyt_drone_amd import evaluate, rollout as R
    n = 256
    venv = P.FixedwingVecEnv(K.train_waypoint_objlock_config(), n, seed=11)
    env = R.VecNormalizeDevice(venv)
    ppo = R.PPO(env, R.PPOConfig(n_steps=8, batch_size=256, n_epochs=2, detector="cnn", image_res=32, seed=5))
    assert isinstance(ppo.policy, R.CnnDetectorPolicy) and not ppo._collect_fused and ppo._fused is None
    w0 = ppo.policy.cnn[0].weight.detach().clone()
    ppo.collect_rollouts()
    img = ppo.buf_img
    assert img.shape == (8, n, 2, 32, 32) and img.is_cuda
    assert float(img[:, :, 1].min()) >= 0.0 and float(img[:, :, 1].max()) <= 1.0 and float(img[:, :, 1].std()) > 1e-3
    assert set(np.unique(img[:, :, 0].cpu().numpy())) <= {0.0, 1.0}
    # the stored image of step t is the render of the state the action of step t was computed from: step 0 = the reset poses
    fresh = P.FixedwingVecEnv(K.train_waypoint_objlock_config(), n, seed=11); fresh.reset_tensor()
    assert torch.equal(fresh.render_tensor(32), img[0])
    ppo.train()
    assert not torch.equal(w0, ppo.policy.cnn[0].weight) and all(np.isfinite(v) for v in ppo.logs.values())
    ppo.learn(2 * 8 * n, reset_num_timesteps=False)
    assert all(bool(torch.isfinite(p).all()) for p in ppo.policy.parameters())
    eval_env = R.VecNormalizeDevice(P.FixedwingVecEnv(K.train_waypoint_objlock_config(), 8, seed=11, global_env_offset=n), training=False, norm_reward=False)
    eval_env.obs_rms.mean.copy_(env.obs_rms.mean); eval_env.obs_rms.var.copy_(env.obs_rms.var)
    res = evaluate.evaluate_policy(ppo.policy, eval_env, n_eval_episodes=8, max_vec_steps=400)
    assert len(res.episode_rewards) >= 1 and all(np.isfinite(res.episode_rewards))
