"""Cost of the CNN front end's inputs and of a CNN-policy training iteration (NOT the headline metric -- bench.py is).
    python tools/bench_render.py [envs] [res]
fw_render launch time at `envs` combined-task envs (20 cylinders) and `res` x `res` pixels; collector and update of
PPO(detector="cnn") with configs[4]'s env (32 768 samples per update, batch 1024)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = int(sys.argv[2]) if len(sys.argv) > 2 else 32
venv = P.FixedwingVecEnv(K.train_waypoint_objlock_config(), n, seed=42)
venv.reset_tensor()
out = torch.empty((n, 2, res, res), dtype=torch.float32, device=venv.device)
for _ in range(5):
    venv.render_tensor(res, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    venv.render_tensor(res, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
line = {"task": "combined", "envs": n, "res": res, "fw_render_us": us, "pixels_per_s": n * res * res / us * 1e6,
        "image_bytes": out.numel() * 4, "write_GBps": out.numel() * 4 / us / 1e3}
if len(sys.argv) > 3 and sys.argv[3] == "render_only":        # (counter passes: fw_render alone)
    print(json.dumps(line)); sys.exit(0)
T = max(32 * 1024 // n, 1)
graphs = not (len(sys.argv) > 3 and sys.argv[3] == "eager")
if os.environ.get("CNN_BENCHMARK"): torch.backends.cudnn.benchmark = True      # MIOpen: search for the fastest algorithm per shape
ppo = R.PPO(R.VecNormalizeDevice(venv), R.PPOConfig(n_steps=T, batch_size=1024, n_epochs=20, detector="cnn", image_res=res, cnn_graphs=graphs))
line["cnn_graphs"] = bool(ppo._graphs)
for _ in range(2):
    ppo.collect_rollouts()
torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    ppo.collect_rollouts()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
line.update(cnn_rollout_us_per_vec_step=dt * 1e6 / (reps * T), cnn_rollout_env_steps_per_s=reps * T * n / dt)
ppo.train(); torch.cuda.synchronize(); t0 = time.perf_counter(); ppo.train(); torch.cuda.synchronize()
line.update(cnn_update_s=time.perf_counter() - t0, update_minibatches=20 * (T * n // 1024))
t0 = time.perf_counter(); ppo.collect_rollouts(); ppo.train(); torch.cuda.synchronize()
line["cnn_end_to_end_env_steps_per_s"] = T * n / (time.perf_counter() - t0)
print(json.dumps(line))
