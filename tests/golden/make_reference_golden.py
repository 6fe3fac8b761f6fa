#!/usr/bin/env python
"""Reference-side fixture generator: the hook that would PIN the oracle to PyFlyt / PyBullet.

CANNOT RUN IN THE BUILD IMAGE: it needs `PyFlyt`, `pybullet`, `gymnasium` (ordinary ModuleNotFoundError there, offline pip) and a
checkout of WdBlink/pyflyt-drone.  Where those exist:

    python tests/golden/make_reference_golden.py --reference /path/to/pyflyt-drone [--cases waypoints objlock combined]

writes `tests/golden/ref_<case>.npz`.  `tests/test_reference_pin.py` is skipped while no `ref_*.npz` is present and otherwise
replays every file on the CPU oracle (same scenario injected through `fw_scenario`, same action trace, motor noise off) and
compares position / attitude at 1e-4 -- BASELINE.json's parity bar.  Until that has run green the oracle is "parity unpinned"
(DESIGN.md section 5 lists the build-owned constants a failing comparison would calibrate).

What it does per case, with the reference's own constructors and keyword values
(train/train_Fixedwing_Waypoints_v3.py:100-117, train/train_objlock.py:113-153, train/train_Fixedwing_Waypoints_ObjLock.py:119-165):
  1. build the env (no SubprocVecEnv: one in-process env), `reset(seed=...)`;
  2. switch the motor noise off on the live drone (best effort: `noise_ratio` of PyFlyt's `Motors`; recorded as `noise_zeroed`);
  3. READ BACK the scenario the reference drew -- waypoints (`WaypointHandler.targets`), duck position (`duck_pos`), obstacle
     bodies (base position from pybullet; height = 2 x base z for a cylinder resting on the ground), and the wind it was
     CONFIGURED with (no randomisation: base / gust / phase are config values, envs/fixedwing_envs/fixedwing_base_env.py:135-171);
  4. replay a fixed seeded action trace; after every step store `Aviary.state(0)` (ang_vel, ang_pos, lin_vel, lin_pos --
     envs/fixedwing_envs/fixedwing_base_env.py:279-285), `aux_state(0)`, the flattened observation, reward, flags.
Only data leaves this script (inputs and outputs of the reference run) -- no reference source."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WIND_FIXED = {"enabled": True, "mode": "gust_sine", "wind_enu_mps": [3.0, -2.0, 0.1], "gust_amp_enu_mps": [1.0, 0.5, 0.05],
              "gust_freq_hz": 0.2, "gust_phase_rad": 0.7, "randomize_on_reset": False, "randomize_gust_phase": False}


def actions_for(case: str, steps: int) -> np.ndarray:
    """The trace both sides replay: smooth, bounded, seeded per case (gentle enough that the flight lasts `steps`)."""
    rng = np.random.default_rng({"waypoints": 11, "objlock": 12, "combined": 13}[case])
    t = np.arange(steps)[:, None] / 30.0
    a = 0.25 * np.sin(2 * np.pi * t * rng.uniform(0.05, 0.4, size=(1, 4)) + rng.uniform(0, 6.28, size=(1, 4)))
    a[:, 3] = 0.4 + 0.3 * np.sin(2 * np.pi * t[:, 0] * 0.07)
    return np.clip(a, -1.0, 1.0)


def zero_motor_noise(base_env) -> bool:
    try:
        drone = base_env.env.drones[0]
        m = getattr(drone, "motors", None)
        if m is not None and hasattr(m, "noise_ratio"):
            m.noise_ratio = np.zeros_like(np.asarray(m.noise_ratio, dtype=np.float64))
            return True
    except Exception:
        pass
    return False


def build(case: str, reference: str):
    sys.path.insert(0, reference)
    import gymnasium as gym
    import PyFlyt.gym_envs  # noqa: F401  (registers PyFlyt/Fixedwing-Waypoints-v3)
    if case == "waypoints":
        from envs.flatten_waypoint_env import FlattenWaypointEnv
        env = gym.make("PyFlyt/Fixedwing-Waypoints-v3", sparse_reward=True, num_targets=8, goal_reach_distance=4, render_mode=None,
                       angle_representation="euler", flight_dome_size=100.0, max_duration_seconds=120.0, agent_hz=30)
        return FlattenWaypointEnv(env, context_length=2), {"wind": None}
    if case == "objlock":
        from envs.fixedwing_objlock_env import FixedwingObjLockEnv
        from envs.flatten_objlock_env import FlattenObjLockEnv
        env = FixedwingObjLockEnv(sparse_reward=False, render_mode="rgb_array", angle_representation="euler", flight_dome_size=200.0,
                                  max_duration_seconds=60.0, agent_hz=30, wind_config=WIND_FIXED, num_obstacles=0,
                                  duck_camera_capture_interval_steps=12, duck_lock_hold_steps=5, duck_strike_distance_m=10.0,
                                  duck_strike_reward=400.0, duck_lock_step_reward=0.2, duck_approach_reward_scale=0.1, duck_global_scaling=60.0)
        return FlattenObjLockEnv(env), {"wind": WIND_FIXED}
    from envs.fixedwing_waypoint_objlock_env import FixedwingWaypointObjLockEnv
    from envs.flatten_waypoint_env import FlattenWaypointEnv
    env = FixedwingWaypointObjLockEnv(sparse_reward=False, num_targets=8, goal_reach_distance=8, flight_dome_size=100.0,
                                      max_duration_seconds=120.0, angle_representation="euler", agent_hz=30, render_mode="rgb_array",
                                      num_obstacles=20, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0), obstacle_safe_distance_m=5.0,
                                      duck_camera_capture_interval_steps=6, duck_lock_hold_steps=10, duck_strike_distance_m=8,
                                      duck_strike_reward=200.0, duck_global_scaling=30.0, wind_config=WIND_FIXED)
    return FlattenWaypointEnv(env, context_length=2), {"wind": WIND_FIXED}


def run_case(case: str, reference: str, steps: int, seed: int) -> str:
    import pybullet as p
    env, meta = build(case, reference)
    obs0, _ = env.reset(seed=seed)
    base = env.unwrapped
    noise_zeroed = zero_motor_noise(base)
    out = {"case": case, "seed": seed, "noise_zeroed": noise_zeroed, "actions": actions_for(case, steps), "obs0": np.asarray(obs0, np.float64)}
    wp = getattr(base, "waypoints", None)
    if wp is not None and hasattr(wp, "targets"):
        out["targets"] = np.asarray(wp.targets, np.float64)
    if getattr(base, "duck_pos", None) is not None:
        out["duck_pos"] = np.asarray(base.duck_pos, np.float64)
    ids = list(getattr(base, "obstacle_ids", []) or [])
    if ids:
        pos = np.array([p.getBasePositionAndOrientation(u, physicsClientId=base.env._client)[0] for u in ids], np.float64)
        out["obstacles"] = np.stack([pos[:, 0], pos[:, 1], 2.0 * pos[:, 2]], axis=1)      # x, y, height (cylinder resting on the ground)
    if meta["wind"]:
        w = meta["wind"]
        out.update(wind_base=np.asarray(w["wind_enu_mps"], np.float64), gust_amp=np.asarray(w["gust_amp_enu_mps"], np.float64),
                   gust_phase=np.float64(w["gust_phase_rad"]), gust_freq_hz=np.float64(w["gust_freq_hz"]))
    st0 = np.asarray(base.env.state(0), np.float64)
    out["state0"] = st0
    states, auxs, obss, rews, terms, truncs = [], [], [], [], [], []
    for a in out["actions"]:
        o, r, te, tr, _ = env.step(a)
        states.append(np.asarray(base.env.state(0), np.float64)); auxs.append(np.asarray(base.env.aux_state(0), np.float64))
        obss.append(np.asarray(o, np.float64)); rews.append(float(r)); terms.append(bool(te)); truncs.append(bool(tr))
        if te or tr:
            break
    out.update(states=np.array(states), aux=np.array(auxs), obs=np.array(obss), rewards=np.array(rews), terminated=np.array(terms),
               truncated=np.array(truncs))
    try:
        import PyFlyt
        out["pyflyt_version"] = str(getattr(PyFlyt, "__version__", "unknown"))
    except Exception:
        out["pyflyt_version"] = "unknown"
    path = os.path.join(HERE, f"ref_{case}.npz")
    np.savez_compressed(path, **out)
    env.close()
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of WdBlink/pyflyt-drone")
    ap.add_argument("--cases", nargs="*", default=["waypoints", "objlock", "combined"])
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    try:
        import gymnasium, pybullet, PyFlyt  # noqa: F401,E401
    except ImportError as e:
        print(f"cannot generate reference fixtures here: {e} (this script needs PyFlyt + pybullet + gymnasium; the oracle stays "
              f"'parity unpinned' until it has run)")
        return 3
    for c in a.cases:
        print("wrote", run_case(c, a.reference, a.steps, a.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
