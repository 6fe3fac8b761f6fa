import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
for name, cfg in (("objlock", K.train_objlock_config()), ("combined", K.train_waypoint_objlock_config())):
    n = 4096
    env = P.FixedwingVecEnv(cfg, n, seed=42); env.reset_tensor()
    g = torch.Generator().manual_seed(0)
    acts = [((torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1) * torch.tensor([0.15, 0.15, 0.15, 0.4], dtype=torch.float64)).cuda() for _ in range(8)]
    def timed(k0, k1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k1 - k0 + 1)]
        ev[0].record()
        for i in range(k0, k1):
            env.step_tensor(acts[i % 8]); ev[i - k0 + 1].record()
        torch.cuda.synchronize()
        return np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(k1 - k0)])
    t_sync = timed(0, 60)            # all envs share the capture phase (fresh reset, no episode has ended yet)
    ends0 = int(env.get_counters()["resets"])
    for i in range(60, 3000): env.step_tensor(acts[i % 8])
    torch.cuda.synchronize()
    t_desync = timed(3000, 3060)
    print(name, "in-sync step times:", " ".join(f"{x:.0f}" for x in t_sync))
    print(f"{name}: eager us per step -- phases in sync (steps 0..59, resets so far {ends0}): mean {t_sync.mean():.1f}, capture steps {np.sort(t_sync)[-20:].mean():.1f}, other steps {np.sort(t_sync)[:40].mean():.1f}; "
          f"desynchronised (steps 3000..3059, resets {env.get_counters()['resets']}): mean {t_desync.mean():.1f} min {t_desync.min():.1f} max {t_desync.max():.1f}")
