"""Dev tool: rate of the host-buffer (numpy) VecEnv.step() surface, PCIe and Python included (DESIGN.md section 6 note)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
env = P.FixedwingWaypointsVecEnv(num_envs=4096, seed=42, sparse_reward=True, num_targets=8, goal_reach_distance=4,
                                 angle_representation="euler", flight_dome_size=100.0, max_duration_seconds=120.0, agent_hz=30, context_length=2)
env.reset()
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (4096, 4)) for _ in range(16)]
for i in range(20): env.step(acts[i % 16])
t0 = time.perf_counter(); n = 200
for i in range(n): env.step(acts[i % 16])
dt = (time.perf_counter() - t0) / n
print(f"numpy VecEnv.step(): {dt*1e6:.0f} us per 4096-env step = {4096/dt/1e6:.2f} M env-steps/s (H2D actions, D2H obs/reward/flags/info, info dicts)")
