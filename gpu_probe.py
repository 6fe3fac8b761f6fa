"""Scratch: first GPU contact (not part of the test-suite)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
from oracle import fw_oracle as O
from helpers import run_lockstep
for dtype, atol in (("float64", 1e-7), ("float32", 5e-2)):
    for noise in (False, True):
        cfg = K.train_waypoints_v3_config(dtype=dtype, motor_noise=noise)
        n = 256
        he = P.FixedwingVecEnv(cfg, n, seed=42); oe = O.OracleEnv(cfg, n, seed=42)
        try:
            w = run_lockstep(he, oe, 300, np.random.default_rng(0), kind="uniform", atol=atol, rtol=1e-6)
            print(dtype, "noise", noise, "OK", w, flush=True)
        except AssertionError as e:
            print(dtype, "noise", noise, "FAIL", str(e)[:1500], flush=True)
# speed
for dtype in ("float64", "float32"):
    for n in (4096, 65536, 1048576):
        cfg = K.train_waypoints_v3_config(dtype=dtype)
        he = P.FixedwingVecEnv(cfg, n, seed=42)
        he.reset_tensor()
        a = torch.rand((n, 4), device=he.device, dtype=he.torch_dtype) * 2 - 1
        for _ in range(20): he.step_tensor(a)
        torch.cuda.synchronize(); t0 = time.perf_counter(); K_ = 200
        for _ in range(K_): he.step_tensor(a)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K_
        print(f"{dtype} N={n}: {dt*1e6:.1f} us/step  {n/dt/1e6:.1f} M env-steps/s", flush=True)
        he.close()
