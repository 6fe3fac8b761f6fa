"""Dev tool: worst deviation of the HIP path from the CPU oracle over lockstep traces (the tests only assert a bound)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
from oracle import fw_oracle as O
from helpers import run_lockstep
O.build()
CASES = {"waypoints (headline config)": K.train_waypoints_v3_config(),
         "waypoints + gust wind (force)": K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND),
         "objlock train config": K.train_objlock_config(),
         "combined train config": K.train_waypoint_objlock_config()}
for lanes in ("8", "1"):
    os.environ["FWSIM_LANES_PER_ENV"] = lanes
    for name, cfg in CASES.items():
        n = 256
        hip, ora = P.FixedwingVecEnv(cfg, n, seed=7), O.OracleEnv(cfg, n, seed=7)
        w = run_lockstep(hip, ora, 240, np.random.default_rng(3), atol=1e-6, rtol=0, state_atol=1e-3)
        print(f"lanes/env={lanes} {name}: 240 steps x {n} envs, {w['dones']} episode ends; worst |obs| {w['obs']:.2e} |reward| {w['rew']:.2e} "
              f"|terminal obs| {w['tobs']:.2e} |state| {w['state']:.2e}", flush=True)
        hip.close()
