"""Dev tool: step time of the lane mappings over env counts (where should fw_create switch from 8 lanes per env at one wave per
SIMD to the 256-register build at two, and from there to one lane per env?).
usage: python tools/crossover.py <waypoints|waypoints_wind|objlock|combined> n1 n2 ..."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
CFG = {"waypoints": K.train_waypoints_v3_config, "objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config,
       "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND)}
which = sys.argv[1]
for n in map(int, sys.argv[2:]):
    modes = ((8, 1), (8, 2), (1, 1)) if which.startswith("waypoints") else ((8, 1), (1, 1))     # (lanes per env, waves per SIMD of the 8-lane build)
    for lanes, waves in modes:
        os.environ["FWSIM_LANES_PER_ENV"] = str(lanes); os.environ["FWSIM_G8_WAVES"] = str(waves)
        e = P.FixedwingVecEnv(CFG[which](), n, seed=42); e.reset_tensor()
        g = torch.Generator().manual_seed(0)
        acts = [(torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(4)]
        for i in range(40): e.step_tensor(acts[i % 4])          # into steady state (captures, resets)
        torch.cuda.synchronize()
        reps = []
        for _ in range(7):                                       # median of 7 x 20 steps (a single timed window once caught an 80 ms hiccup)
            t0 = time.perf_counter()
            for i in range(20): e.step_tensor(acts[i % 4])
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 20)
        dt = sorted(reps)[len(reps) // 2]
        print(f"{which} N={n} lanes={lanes} waves/SIMD={waves if lanes == 8 else 2}: {dt*1e6:.1f} us/step  {n/dt/1e6:.1f} M env-steps/s", flush=True)
        e.close(); del e
