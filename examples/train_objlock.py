"""Device-resident counterpart of the reference's train/train_objlock.py (:27-86 config, :113-153 env, :188-298 train()).

Same env parameters (dome 200 m, 60 s, duck scale 60, hold 5 / strike 10 m / +400, gust wind ranges, no obstacles,
480 x 480 camera via render_mode="rgb_array"), same PPO hyper-parameters (batch 64, 10 epochs, lr 3e-4, ent 0.001); the env
count goes from 16 to thousands and n_steps shrinks so that one update still sees 16 x 2048 = 32 768 samples.  The eval
callback reports the reference's ObjLock metrics: success_rate (is_success = strike) and duck_strike_rate.

    python examples/train_objlock.py --total_timesteps 2000000 --num_envs 4096 --out runs/objlock
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P  # noqa: E402
from pyflyt_drone_amd import checkpoint, evaluate, rollout as R  # noqa: E402

TRAIN_CONFIG = dict(total_timesteps=1_000_000, n_eval_episodes=10, learning_rate=3e-4, samples_per_update=16 * 2048, batch_size=64,
                    n_epochs=10, gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.001, vf_coef=0.5, max_grad_norm=0.5, seed=42)
WIND = {"enabled": True, "mode": "gust_sine", "wind_enu_mps": [0.0, 0.0, 0.0],
        "wind_enu_mps_range": [[-10.0, 10.0], [-10.0, 10.0], [-0.10, 0.10]], "gust_amp_enu_mps": [0.0, 0.0, 0.0],
        "gust_amp_enu_mps_range": [[0.0, 3.0], [0.0, 3.0], [0.0, 0.3]], "gust_freq_hz": 0.2, "gust_phase_rad": 0.0,
        "randomize_on_reset": True, "randomize_gust_phase": True}
ENV_KW = dict(sparse_reward=False, flight_dome_size=200.0, max_duration_seconds=60.0, angle_representation="euler", agent_hz=30,
              render_mode="rgb_array", num_obstacles=0, obstacle_radius=2.0, obstacle_height_range=(10.0, 30.0),
              obstacle_safe_distance_m=10.0, obstacle_avoid_reward_scale=1.0, obstacle_avoid_max_penalty=5.0,
              duck_camera_capture_interval_steps=12, duck_lock_hold_steps=5, duck_strike_distance_m=10.0, duck_strike_reward=400.0,
              duck_lock_step_reward=0.2, duck_approach_reward_scale=0.1, duck_global_scaling=60.0, wind_config=WIND)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model", type=str, default=None)
    ap.add_argument("--vecnorm_path", type=str, default=None)
    ap.add_argument("--total_timesteps", type=int, default=None)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--out", type=str, default="runs/obj_lock_only_ppo")
    a = ap.parse_args()
    cfg = TRAIN_CONFIG
    model_dir, log_dir = os.path.join(a.out, "models"), os.path.join(a.out, "logs")
    os.makedirs(model_dir, exist_ok=True); os.makedirs(log_dir, exist_ok=True)

    world, rank, local = R.init_distributed_from_env()          # torchrun: one process per GPU (RCCL); (1, 0, 0) otherwise
    dev = local if world > 1 else None
    # rank r simulates global envs [r * num_envs, (r + 1) * num_envs): scenario / noise / sampling streams are keyed on the global id
    env = R.VecNormalizeDevice(P.FixedwingObjLockVecEnv(num_envs=a.num_envs, seed=cfg["seed"], device=dev, global_env_offset=rank * a.num_envs, **ENV_KW))
    eval_env = R.VecNormalizeDevice(P.FixedwingObjLockVecEnv(num_envs=16, seed=cfg["seed"], device=dev, global_env_offset=world * a.num_envs, **ENV_KW),
                                    training=False, norm_reward=False)
    vecnorm = checkpoint.infer_vecnorm_path(a.pretrained_model, a.vecnorm_path, model_dir)
    if vecnorm:
        checkpoint.load_vecnormalize(vecnorm, env, training=True, norm_reward=True)
    n_steps = R.n_steps_for(cfg["samples_per_update"], a.num_envs, world)      # holds the samples per update: n_steps ~ 1 / (envs x world)
    model = R.PPO(env, R.PPOConfig(n_steps=n_steps, batch_size=cfg["batch_size"], n_epochs=cfg["n_epochs"], learning_rate=cfg["learning_rate"],
                                   gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"], clip_range=cfg["clip_range"], ent_coef=cfg["ent_coef"],
                                   vf_coef=cfg["vf_coef"], max_grad_norm=cfg["max_grad_norm"], seed=cfg["seed"]))
    if a.pretrained_model:
        checkpoint.set_parameters(a.pretrained_model, model)
    ev = evaluate.EvalCallback(eval_env, n_eval_episodes=max(cfg["n_eval_episodes"], 16), eval_freq=max(10000 // a.num_envs, 1) * 50,
                               log_path=log_dir, best_model_save_path=model_dir, verbose=1)
    ck = checkpoint.CheckpointCallback(save_freq=max(50000 // a.num_envs, 1) * 50, save_path=model_dir, name_prefix="obj_lock_ppo")

    class Progress:
        t0, last = time.perf_counter(), 0
        def on_rollout_end(self, ppo):
            if ppo.num_timesteps - self.last >= 20 * n_steps * a.num_envs * world:
                dt = time.perf_counter() - self.t0
                print(json.dumps({"timesteps": ppo.num_timesteps, "fps": round(ppo.num_timesteps / dt), **{k: round(v, 5) for k, v in ppo.logs.items()},
                                  **{k: round(float(v), 4) for k, v in ev.last_scalars.items()}}), flush=True)
                self.last = ppo.num_timesteps
            return True

    total = a.total_timesteps if a.total_timesteps is not None else cfg["total_timesteps"]
    try:
        model.learn(total, callbacks=[ev, ck, Progress()], reset_num_timesteps=True)
    finally:
        if model.rank == 0:                 # one writer per file in a multi-process job
            checkpoint.save(os.path.join(model_dir, "final_model.pt"), model)
            checkpoint.save_vecnormalize(os.path.join(model_dir, "vecnorm.pt"), env)
        env.venv.close(); eval_env.venv.close()


if __name__ == "__main__":
    main()
