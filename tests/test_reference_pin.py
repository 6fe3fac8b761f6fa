"""The pin of oracle == reference physics: replays `tests/golden/ref_*.npz` -- traces of the reference itself (PyFlyt / PyBullet),
made by `tests/golden/make_reference_golden.py` where those packages exist -- on the CPU oracle with the same scenario injected
(`fw_scenario`), the same actions and motor noise off, and compares position / attitude at BASELINE.json's 1e-4.
Skipped while no reference fixture is present (the build image cannot produce one: PyFlyt / pybullet are not installable
there); until it runs green, parity against PyBullet is UNPINNED and every document says so."""
import glob
import os

import numpy as np
import pytest

from pyflyt_drone_amd import config as K

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_*.npz")))
CONFIGS = {"waypoints": lambda: K.train_waypoints_v3_config(motor_noise=False),
           "objlock": lambda: K.train_objlock_config(motor_noise=False),
           "combined": lambda: K.train_waypoint_objlock_config(motor_noise=False)}


def test_fixture_generator_is_committed_and_refuses_politely_without_pyflyt():
    """The hook exists and says what it needs (exit code 3 where the reference's dependencies do not import)."""
    import subprocess, sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_reference_golden.py")
    assert os.path.exists(script)
    try:
        import PyFlyt  # noqa: F401
        pytest.skip("PyFlyt imports here: run the generator instead")
    except ImportError:
        pass
    p = subprocess.run([sys.executable, script, "--reference", "/nonexistent"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "parity unpinned" in p.stdout


@pytest.mark.skipif(not FIXTURES, reason="no tests/golden/ref_*.npz: reference fixtures cannot be generated in this image (PyFlyt / pybullet absent) -- parity vs PyBullet unpinned")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_follows_the_reference_trace_within_1e_4(oracle, path):
    z = np.load(path, allow_pickle=False)
    case = str(z["case"])
    cfg = CONFIGS[case]()
    if "wind_base" in z.files:                   # the reference run used a fixed, non-random wind
        cfg.wind_randomize_on_reset = 0; cfg.wind_randomize_phase = 0
        cfg.gust_freq_hz = float(z["gust_freq_hz"])
    env = oracle.OracleEnv(cfg, 1, seed=int(z["seed"]))
    kw = {}
    if "targets" in z.files:
        kw["targets"] = z["targets"][None, :K.FW_MAX_TARGETS, :3]
    if "duck_pos" in z.files:
        kw["duck_pos"] = z["duck_pos"].reshape(1, 3)
    if "obstacles" in z.files:
        k = min(len(z["obstacles"]), K.FW_MAX_OBSTACLES)
        kw["obstacles"] = z["obstacles"][None, :k]; kw["num_obstacles"] = np.array([k])
    if "wind_base" in z.files:
        kw.update(wind_base=z["wind_base"].reshape(1, 3), gust_amp=z["gust_amp"].reshape(1, 3), gust_phase=np.array([float(z["gust_phase"])]))
    sc, keep = K.make_scenario(1, **kw)
    obs0 = env.reset(scenario=sc)
    # the state after reset (10 warm-up Aviary steps) is the first thing that must agree: rows of Aviary.state(0) =
    # [ang_vel(body), ang_pos(euler), lin_vel(body), lin_pos(world)] (envs/fixedwing_envs/fixedwing_base_env.py:279-285) = obs[0:12]
    np.testing.assert_allclose(obs0[0, :12], z["state0"].reshape(-1), rtol=0, atol=1e-4, err_msg="state after reset")
    for t, a in enumerate(z["actions"][:len(z["states"])]):
        obs, rew, te, tr, _, _ = env.step(a.reshape(1, 4))
        want = z["states"][t].reshape(-1)
        np.testing.assert_allclose(obs[0, 9:12], want[9:12], rtol=0, atol=1e-4, err_msg=f"position, step {t}")
        np.testing.assert_allclose(obs[0, 3:6], want[3:6], rtol=0, atol=1e-4, err_msg=f"attitude, step {t}")
        assert bool(te[0]) == bool(z["terminated"][t]) and bool(tr[0]) == bool(z["truncated"][t]), f"flags, step {t}"
