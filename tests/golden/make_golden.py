"""Generates the committed golden vectors with the build's own CPU oracle.

They pin  HIP kernel == oracle  and guard the oracle against regressions; they do
NOT pin oracle == PyBullet (the reference cannot run here: PyFlyt / pybullet /
gymnasium are absent and the reference ships no fixtures -- SURVEY.md section 8c).

    python tests/golden/make_golden.py [name ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import pyflyt_drone_amd  # noqa: E402,F401
from pyflyt_drone_amd import config as K  # noqa: E402
from oracle import fw_oracle as O  # noqa: E402
from helpers import seeded_actions  # noqa: E402

GUST = dict(enabled=True, mode="gust_sine", randomize_on_reset=True, randomize_gust_phase=True,
            wind_enu_mps_range=[[-5, 5], [-5, 5], [-0.5, 0.5]], gust_amp_enu_mps_range=[[0, 3], [0, 3], [0, 0.3]],
            gust_freq_hz=0.2)                      # train/train_Fixedwing_Waypoints_ObjLock.py:59-70
CONST_AIR = dict(enabled=True, mode="constant", wind_enu_mps=[2.0, -3.0, 0.25], coupling="airspeed")


def cases():
    yield "train_v3_sparse_euler_noise", K.train_waypoints_v3_config(), "uniform", 4, 100, 42
    yield "dense_quat_ctx3_gust_force", K.waypoints_config(
        sparse_reward=False, num_targets=3, goal_reach_distance=8.0, angle_representation="quaternion",
        context_length=3, wind_config=GUST), "gentle", 4, 100, 7
    yield "dense_euler_const_airspeed_nonoise", K.waypoints_config(
        sparse_reward=False, num_targets=2, goal_reach_distance=6.0, angle_representation="euler",
        context_length=1, wind_config=CONST_AIR, motor_noise=False, max_duration_seconds=2.0), "gentle", 4, 100, 11
    yield "objlock_train_config", K.train_objlock_config(duck_camera_capture_interval_steps=3, num_obstacles=6,
                                                         obstacle_safe_distance_m=60.0), "gentle", 4, 100, 21
    yield "combined_train_config", K.train_waypoint_objlock_config(goal_reach_distance=30.0, duck_camera_capture_interval_steps=2,
                                                                   obstacle_safe_distance_m=40.0), "gentle", 4, 100, 31
    # duck_vision_use_deltas=False (envs/fixedwing_objlock_env.py:69-70, 440-441): the observation ends with the history, 52 wide
    nd = K.train_objlock_config(duck_camera_capture_interval_steps=2)
    nd.duck_vision_no_deltas = 1
    yield "objlock_no_deltas", nd, "gentle", 4, 100, 41


def main(only=None):
    """`python tests/golden/make_golden.py [name ...]`: all cases, or only the named ones (the others' files stay as committed)."""
    O.build()
    for name, cfg, kind, n, steps, seed in cases():
        if only and name not in only:
            continue
        env = O.OracleEnv(cfg, n, seed=seed)
        rng = np.random.default_rng(seed)
        obs0 = env.reset()
        A, OB, R, TE, TR, TO, IN = [], [], [], [], [], [], []
        for _ in range(steps):
            a = seeded_actions(rng, n, kind)
            o, r, te, tr, to, info = env.step(a)
            A.append(a); OB.append(o); R.append(r); TE.append(te); TR.append(tr); TO.append(to); IN.append(info)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, seed=seed, obs0=obs0, actions=np.array(A), obs=np.array(OB), reward=np.array(R),
                            terminated=np.array(TE), truncated=np.array(TR), terminal_obs=np.array(TO),
                            info=np.array(IN), final_state=env.get_state())
        done = int((np.array(TE) | np.array(TR)).sum())
        print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, {done} episode ends")


if __name__ == "__main__":
    main(sys.argv[1:])
