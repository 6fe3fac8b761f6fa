"""Device-resident counterpart of the reference's train/train_Fixedwing_Waypoints_v3.py (:236-347).

Same TRAIN_CONFIG (:27-55) with the env count raised from 32 to thousands and n_steps lowered so that one
update still sees 32 x 2048 = 65 536 samples; same callbacks (eval every ~10 000 steps with the waypoint reach
rates, periodic checkpoints, best model), same final artefacts (final_model + vecnorm), same restart semantics
(--pretrained_model loads parameters only and the step counter starts from zero).

    python examples/train_fixedwing_waypoints.py --total_timesteps 2000000 --num_envs 4096 --out runs/wp
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P  # noqa: E402
from pyflyt_drone_amd import checkpoint, evaluate, rollout as R  # noqa: E402

TRAIN_CONFIG = dict(seed=42, total_timesteps=20_000_000, n_eval_episodes=5, learning_rate=3e-4, batch_size=128, n_epochs=20,
                    gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.001, vf_coef=0.5, max_grad_norm=0.5,
                    samples_per_update=32 * 2048)
ENV_KW = dict(sparse_reward=True, num_targets=8, goal_reach_distance=4, angle_representation="euler", flight_dome_size=100.0,
              max_duration_seconds=120.0, agent_hz=30, context_length=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model", type=str, default=None)
    ap.add_argument("--vecnorm_path", type=str, default=None)
    ap.add_argument("--total_timesteps", type=int, default=None)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--out", type=str, default="runs/waypoints_ppo")
    a = ap.parse_args()
    cfg = TRAIN_CONFIG
    model_dir, log_dir = os.path.join(a.out, "models"), os.path.join(a.out, "logs")
    os.makedirs(model_dir, exist_ok=True); os.makedirs(log_dir, exist_ok=True)

    world, rank, local = R.init_distributed_from_env()          # torchrun: one process per GPU (RCCL); (1, 0, 0) otherwise
    dev = local if world > 1 else None
    # rank r simulates global envs [r * num_envs, (r + 1) * num_envs): scenario / noise / sampling streams are keyed on the global id
    env = R.VecNormalizeDevice(P.FixedwingWaypointsVecEnv(num_envs=a.num_envs, seed=cfg["seed"], device=dev, global_env_offset=rank * a.num_envs, **ENV_KW))
    eval_env = R.VecNormalizeDevice(P.FixedwingWaypointsVecEnv(num_envs=16, seed=cfg["seed"], device=dev, global_env_offset=world * a.num_envs, **ENV_KW),
                                    training=False, norm_reward=False)
    vecnorm = checkpoint.infer_vecnorm_path(a.pretrained_model, a.vecnorm_path, model_dir)
    if vecnorm:
        checkpoint.load_vecnormalize(vecnorm, env, training=True, norm_reward=True)
    n_steps = R.n_steps_for(cfg["samples_per_update"], a.num_envs, world)      # holds the samples per update: n_steps ~ 1 / (envs x world)
    model = R.PPO(env, R.PPOConfig(n_steps=n_steps, batch_size=cfg["batch_size"], n_epochs=cfg["n_epochs"],
                                   learning_rate=cfg["learning_rate"], gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"],
                                   clip_range=cfg["clip_range"], ent_coef=cfg["ent_coef"], vf_coef=cfg["vf_coef"],
                                   max_grad_norm=cfg["max_grad_norm"], seed=cfg["seed"]))
    if a.pretrained_model:
        checkpoint.set_parameters(a.pretrained_model, model)
    ev = evaluate.EvalCallback(eval_env, n_eval_episodes=max(cfg["n_eval_episodes"], 16), eval_freq=max(10000 // a.num_envs, 1) * 50,
                               log_path=log_dir, best_model_save_path=model_dir, num_targets_total=ENV_KW["num_targets"], verbose=1)
    ck = checkpoint.CheckpointCallback(save_freq=max(50000 // a.num_envs, 1) * 50, save_path=model_dir, name_prefix="waypoints_ppo")

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
