"""Dev tool: soak runs (long step sequences at sizes the tests do not reach) -- looks for hangs, NaNs, stuck episodes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R
t0 = time.time()
for name, cfg, n, steps in (("objlock g1", K.train_objlock_config(), 65536, 1500),
                            ("combined, 8-lane mapping beyond 16 384 envs (cylinders)", K.train_waypoint_objlock_config(), 32768, 1500),
                            ("waypoints_wind g1", K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND), 131072, 1500),
                            ("waypoints g8 odd n", K.train_waypoints_v3_config(), 4099, 5000),
                            ("combined g8 (wave-level camera), largest latency-mapped n", K.train_waypoint_objlock_config(), 16384, 2500),
                            ("combined g8 odd n", K.train_waypoint_objlock_config(duck_camera_capture_interval_steps=1), 4099, 2500),
                            ("objlock + obstacles g8 odd n", K.train_objlock_config(num_obstacles=20, duck_camera_capture_interval_steps=2), 4099, 2500)):
    e = P.FixedwingVecEnv(cfg, n, seed=3); e.reset_tensor()
    g = torch.Generator().manual_seed(1)
    acts = [(torch.rand((n, 4), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(8)]
    done = 0
    for i in range(steps):
        e.step_tensor(acts[i % 8])
        if i % 250 == 249:
            done += int((e.terminated | e.truncated).sum())
            assert torch.isfinite(e.obs).all() and torch.isfinite(e.rewards).all(), name
    torch.cuda.synchronize()
    ep = e.get_state()[:, K.S_EPISODE]
    print(f"{name}: n={n} steps={steps} ok; episodes per env min {ep.min():.0f} mean {ep.mean():.1f} max {ep.max():.0f}; t={time.time()-t0:.0f}s", flush=True)
    e.close()
# learner soak: combined env, reference hyper-parameters, ~45 s
env = R.VecNormalizeDevice(P.FixedwingVecEnv(K.train_waypoint_objlock_config(), 4096, seed=5))
ppo = R.PPO(env, R.PPOConfig(n_steps=8, batch_size=128, n_epochs=20, seed=5))
t1 = time.time()
while time.time() - t1 < 45:
    ppo.collect_rollouts(); ppo.train()
assert all(torch.isfinite(p).all() for p in ppo.policy.parameters())
print(f"learner soak (combined): {ppo.num_timesteps} timesteps in {time.time()-t1:.0f}s, logs {ppo.logs}", flush=True)
