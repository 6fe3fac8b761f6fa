"""Dev tool: run 100 physics-only steps of a task config (for rocprofv3)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
which = sys.argv[1]
cfg = {"waypoints": K.train_waypoints_v3_config, "objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config,
       "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND)}[which]()
e = P.FixedwingVecEnv(cfg, 4096, seed=42); e.reset_tensor()
g = torch.Generator().manual_seed(0)
acts = [(torch.rand((4096, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(16)]
for i in range(100): e.step_tensor(acts[i % 16])
torch.cuda.synchronize()
