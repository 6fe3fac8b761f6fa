"""Dev tool: headline step time with an alternative build of the library (A/B experiments).
usage: python tools/bench_lib.py tools/_build/libfwsim_x.so [waypoints|waypoints_wind|objlock|combined]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, _lib
if len(sys.argv) > 1 and sys.argv[1] != "-": _lib.LIB_PATH = os.path.abspath(sys.argv[1])
CFG = {"waypoints": K.train_waypoints_v3_config, "objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config,
       "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND),
       "combined_noobst": lambda: K.train_waypoint_objlock_config(num_obstacles=0),
       "combined_nocam": lambda: K.train_waypoint_objlock_config(duck_camera_capture_interval_steps=10 ** 6),
       "combined_noobst_nocam": lambda: K.train_waypoint_objlock_config(num_obstacles=0, duck_camera_capture_interval_steps=10 ** 6),
       "objlock_nocam": lambda: K.train_objlock_config(duck_camera_capture_interval_steps=10 ** 6)}
N = int(os.environ.get("N", 4096))
for which in sys.argv[2:] or ["waypoints"]:
    e = P.FixedwingVecEnv(CFG[which](), N, seed=42); e.reset_tensor()
    g = torch.Generator().manual_seed(0)
    acts = [(torch.rand((N, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(64)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(320): e.step_tensor(acts[i % 64])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for a in acts: e.step_tensor(a)
        for _ in range(5): gr.replay()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(10): gr.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 640)
    print(f"{os.path.basename(_lib.LIB_PATH)} {which} N={N}: {best*1e6:.2f} us/step  {N/best/1e6:.1f} M env-steps/s", flush=True)
    del e
