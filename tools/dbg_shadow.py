"""Dev tool: steady-state fw_step time (hipGraph of 64 launches) with and without the background warm-up."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
CFG = {"objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config,
       "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND)}
N = int(os.environ.get("N", 4096))
for which in sys.argv[1:] or list(CFG):
    cfg = CFG[which]()
    for shadow in (True, False):
        if not shadow: os.environ["FWSIM_NO_SHADOW"] = "1"
        e = P.FixedwingVecEnv(cfg, N, seed=42); e.reset_tensor()
        os.environ.pop("FWSIM_NO_SHADOW", None)
        g = torch.Generator().manual_seed(0)
        acts = [(torch.rand((N, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(64)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(320): e.step_tensor(acts[i % 64])
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for a in acts: e.step_tensor(a)
            for _ in range(3): gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): gr.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 640
        print(f"{which} N={N} shadow={shadow}: {dt*1e6:.1f} us/step  {N/dt/1e6:.1f} M env-steps/s", flush=True)
        del e
