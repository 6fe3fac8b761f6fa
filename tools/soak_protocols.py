"""Dev tool: soak of the in-grid protocols (fw_collect_step / fw_collect_close / fw_ppo_update) -- thousands of launches per
(task, env count), hipGraph replays, the status words read after every chunk of rollouts; any wait that ran out raises
(rollout.PPO.check_collect_status, FusedPpoUpdate.run).  Sizes: ragged env counts, the two-waves-per-SIMD build, more than 1024
step workgroups, the per-GPU shares of BASELINE.json's configs.

    python tools/soak_protocols.py [seconds per case, default 12]
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
CASES = (("waypoints", K.train_waypoints_v3_config, 4096, dict(n_steps=16, batch_size=128, n_epochs=2)),
         ("waypoints", K.train_waypoints_v3_config, 4099, dict(n_steps=4, batch_size=64, n_epochs=1)),
         ("waypoints (two waves per SIMD)", K.train_waypoints_v3_config, 16384, dict(n_steps=4, batch_size=256, n_epochs=1)),
         ("waypoints + wind", lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND), 4096, dict(n_steps=8, batch_size=128, n_epochs=1)),
         ("objlock", K.train_objlock_config, 4096, dict(n_steps=8, batch_size=64, n_epochs=2)),
         ("combined (configs[4] per-GPU share)", K.train_waypoint_objlock_config, 2048, dict(n_steps=8, batch_size=128, n_epochs=2)),
         ("combined (2048 step workgroups)", K.train_waypoint_objlock_config, 16384, dict(n_steps=4, batch_size=128, n_epochs=1)))
for name, cfg, n, hp in CASES:
    env = R.VecNormalizeDevice(P.FixedwingVecEnv(cfg(), n, seed=11))
    ppo = R.PPO(env, R.PPOConfig(seed=11, **hp))
    assert ppo._one_launch, name
    t0, rollouts, updates = time.time(), 0, 0
    while time.time() - t0 < budget:
        for _ in range(25):
            ppo.collect_rollouts(); rollouts += 1
        ppo.check_collect_status()                  # raises if any wait of any launch so far ran out
        ppo.train(); updates += 1                   # raises if a workgroup of fw_ppo_update gave up
    torch.cuda.synchronize()
    T = hp["n_steps"]
    want = 1e-4 + (rollouts * T + 1) * n
    got = float(env.obs_rms.count)
    assert abs(got - want) < 0.5, (name, got, want)
    assert all(torch.isfinite(p).all() for p in ppo.policy.parameters()) and torch.isfinite(ppo.buf_obs).all()
    assert ppo._one_launch and ppo.collect_fallbacks == 0, (name, "the object fell back to the three-launch collector: a status word was raised")
    print(f"{name}: {n} envs, {rollouts} rollouts = {rollouts * T} fw_collect_step launches + {rollouts} closing launches, {updates} updates "
          f"({updates * hp['n_epochs'] * (T * n // hp['batch_size'])} minibatches, L2 paths {ppo._fused.last_paths:#x}); status 0 throughout, "
          f"{got:.0f} samples in the observation statistics (= expected); {time.time() - t0:.0f} s", flush=True)
    env.venv.close()
