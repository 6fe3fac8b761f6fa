"""Dev tool: soak of the two-wave camera kernels (FWSIM_CAPTURE_WAVE=1) next to the one-wave kernels on the same seeds and
actions: thousands of launches at sizes / settings the tests do not reach, states compared at the end, no wait ever gave up."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
t0 = time.time()
total = 0
for name, cfg, n, steps in (("objlock (training config)", K.train_objlock_config(), 4096, 6000),
                            ("combined (training config)", K.train_waypoint_objlock_config(), 4096, 4000),
                            ("combined, camera every sub-step, odd n", K.train_waypoint_objlock_config(duck_camera_capture_interval_steps=1), 4099, 1500),
                            ("objlock + 20 obstacles, camera every 2nd sub-step, odd n", K.train_objlock_config(num_obstacles=20, duck_camera_capture_interval_steps=2), 4099, 1500),
                            ("combined, 16 384 envs", K.train_waypoint_objlock_config(), 16384, 1000)):
    envs = []
    for cw in ("0", "1"):
        os.environ["FWSIM_CAPTURE_WAVE"] = cw
        e = P.FixedwingVecEnv(cfg, n, seed=3); e.reset_tensor(); envs.append(e)
    assert not envs[0].capture_wave and envs[1].capture_wave
    g = torch.Generator().manual_seed(1)
    acts = [(torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() * 0.5 for _ in range(8)]
    worst = 0.0
    for i in range(steps):
        for e in envs: e.step_tensor(acts[i % 8])
        if i % 250 == 249:
            a, b = envs[0], envs[1]
            assert torch.isfinite(b.obs).all() and torch.isfinite(b.rewards).all(), name
            assert torch.equal(a.terminated, b.terminated) and torch.equal(a.truncated, b.truncated), (name, i)
            worst = max(worst, float((a.obs - b.obs).abs().max()), float((a.rewards - b.rewards).abs().max()))
    torch.cuda.synchronize()
    sa, sb = envs[0].get_state(), envs[1].get_state()
    c = envs[1].get_counters()
    assert c["capture_wave_timeouts"] == 0, c
    total += steps
    print(f"{name}: n={n} launches={steps}: dones equal at every check, max |obs, reward difference| {worst:.2e}, max |state difference| {np.abs(sa - sb).max():.2e}; "
          f"resets {c['resets']} (shadow {c['shadow_hits']}, in-kernel {c['fallbacks']}), capture-wave timeouts {c['capture_wave_timeouts']}; t={time.time()-t0:.0f}s", flush=True)
    for e in envs: e.close()
print(f"{total} launches of the two-wave kernels, no wait gave up")
