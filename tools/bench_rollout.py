"""Collector throughput (not the headline metric): env-steps/s of the full on-device rollout
loop -- policy forward + clip + fw_step + fw_normalize_obs + buffer writes -- and of one
complete PPO iteration, 4096 envs, hyper-parameters of train_Fixedwing_Waypoints_v3.py."""
import json, sys, time
import torch
sys.path.insert(0, ".")
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R

n = 4096
env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), n, seed=42)
vn = R.VecNormalizeDevice(env)
ppo = R.PPO(vn, R.PPOConfig(n_steps=16, batch_size=128, n_epochs=20))
ppo.collect_rollouts(); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 20
for _ in range(reps):
    ppo.collect_rollouts()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = {"rollout_env_steps_per_s": reps * 16 * n / dt, "rollout_ms_per_vec_step": dt * 1e3 / (reps * 16)}
t0 = time.perf_counter(); ppo.train(); torch.cuda.synchronize(); out["train_s_per_update_65536_samples_20_epochs"] = time.perf_counter() - t0
t0 = time.perf_counter(); ppo.collect_rollouts(); ppo.train(); torch.cuda.synchronize()
out["end_to_end_env_steps_per_s"] = 16 * n / (time.perf_counter() - t0)
print(json.dumps(out))
