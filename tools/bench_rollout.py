"""Collector throughput (NOT the headline metric -- bench.py is): env-steps/s of the full
on-device rollout loop (policy forward + clip + fw_step + fw_normalize_obs + buffer writes,
replayed as one hipGraph) and of complete PPO iterations, 4096 envs.

    python tools/bench_rollout.py [waypoints|objlock|combined]

configs[2] of BASELINE.json = objlock (train/train_objlock.py hyper-parameters: batch 64,
10 epochs); waypoints / combined use train_Fixedwing_Waypoints_v3.py / ..._ObjLock.py
(batch 128, 20 epochs).  n_steps is scaled so that one update sees the reference's sample count."""
import json, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R

task = sys.argv[1] if len(sys.argv) > 1 else "waypoints"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096          # envs on this GPU (configs[4] of BASELINE.json: 2048 per GPU)
one = not (len(sys.argv) > 3 and sys.argv[3] == "three")     # PPOConfig.one_launch_collect (fw_collect_step, the default) | "three": the three-launch collector
pmc = len(sys.argv) > 3 and sys.argv[3] == "pmc"             # counter passes (tools/collect_learner_pmc.sh): eager launches (one dispatch record each), few repeats
if task == "objlock":
    cfg, ppo_cfg = K.train_objlock_config(), R.PPOConfig(n_steps=8, batch_size=64, n_epochs=10, one_launch_collect=one, use_graphs=not pmc)        # 16 x 2048 = 32768 samples
elif task == "combined":
    cfg, ppo_cfg = K.train_waypoint_objlock_config(), R.PPOConfig(n_steps=8, batch_size=128, n_epochs=20, one_launch_collect=one, use_graphs=not pmc)  # 32 x 1024 = 32768
else:
    cfg, ppo_cfg = K.train_waypoints_v3_config(), R.PPOConfig(n_steps=16, batch_size=128, n_epochs=20, one_launch_collect=one, use_graphs=not pmc)     # 32 x 2048 = 65536
env = P.FixedwingVecEnv(cfg, n, seed=42)
vn = R.VecNormalizeDevice(env)
ppo = R.PPO(vn, ppo_cfg)
for _ in range(3):
    ppo.collect_rollouts()
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 3 if pmc else 20
for _ in range(reps):
    ppo.collect_rollouts()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
T = ppo_cfg.n_steps
out = {"task": task, "envs": n, "one_launch_collect": bool(ppo._one_launch), "obs_dim": env.obs_dim, "rollout_env_steps_per_s": reps * T * n / dt,
       "rollout_us_per_vec_step": dt * 1e6 / (reps * T)}
ppo.train(); torch.cuda.synchronize()
t0 = time.perf_counter(); ppo.train(); torch.cuda.synchronize()
out["update_s"] = time.perf_counter() - t0
out["update_minibatches"] = ppo_cfg.n_epochs * (T * n // ppo_cfg.batch_size)
out["update_paths"] = ppo._fused.last_paths if ppo._fused is not None else None      # fw_ppo_update_status: which exchanges shared an L2
out["us_per_minibatch"] = out["update_s"] * 1e6 / out["update_minibatches"]
if pmc:
    print(json.dumps(out)); sys.exit(0)
t0 = time.perf_counter(); ppo.collect_rollouts(); ppo.train(); torch.cuda.synchronize()
out["end_to_end_env_steps_per_s_reference_hparams"] = T * n / (time.perf_counter() - t0)
# the same sample count with minibatches sized for a GPU (one of the two knobs a user turns first)
big = R.PPO(R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, n, seed=43)), R.PPOConfig(n_steps=T, batch_size=4096, n_epochs=ppo_cfg.n_epochs))
for _ in range(2):
    big.collect_rollouts(); big.train()
torch.cuda.synchronize(); t0 = time.perf_counter(); big.collect_rollouts(); big.train(); torch.cuda.synchronize()
out["end_to_end_env_steps_per_s_batch4096"] = T * n / (time.perf_counter() - t0)
print(json.dumps(out))
