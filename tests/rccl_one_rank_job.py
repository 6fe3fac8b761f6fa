"""One rank, RCCL, every collective of the sharded path on device tensors (started by tests/test_sharded_training_gpu.py through
the launcher helper with FW_DIST_FORCE=1 and a one-rank rendezvous in the environment).

RCCL refuses two ranks on one device, so on a one-GPU box the two-rank rehearsals run over gloo and take the host-copy branches of
rollout.all_reduce_sum_ / all_gather_cat / the status agreement.  This job is the other half: the `backend == "nccl"` branches --
broadcast of the initial weights, the statistics all-reduce, all_gather_into_tensor of the rollout shard, the gradient all-reduce
of dist_update="allreduce", the collective status look of train(), the evaluation broadcast -- on the real library, with the
arithmetic of a single-process job to compare against.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pyflyt_drone_amd as P  # noqa: E402
from pyflyt_drone_amd import config as K  # noqa: E402
from pyflyt_drone_amd import evaluate as E  # noqa: E402
from pyflyt_drone_amd import rollout as R  # noqa: E402


def run(dist_update, forced):
    cfg = K.train_waypoints_v3_config()
    env = R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, 256, device=0, seed=42))
    ppo = R.PPO(env, R.PPOConfig(n_steps=8, batch_size=128, n_epochs=2, seed=42, dist_update=dist_update))
    for _ in range(2):
        ppo.collect_rollouts(); ppo.train()
    flat = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).double()
    out = dict(sharded=R._dist() is not None, replicated=bool(ppo._replicated), one_launch=bool(ppo._one_launch), graphs=bool(ppo._graphs),
               allgather_bytes=float(ppo.allgather_bytes), checksum=float(ppo.replica_checksum()), fallbacks=int(ppo.collect_fallbacks),
               weights_sum=float(flat.sum()), weights_abs=float(flat.abs().sum()), obs_mean=float(env.obs_rms.mean.sum()),
               count=float(env.obs_rms.count.sum()), timesteps=int(ppo.num_timesteps), finite=bool(torch.isfinite(flat).all()))
    if forced and dist_update == "replicated":
        # the evaluation record's broadcast of rank 0's figure (evaluate.EvalCallback._record), on a device tensor
        ev = R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, 8, device=0, seed=7), training=False, norm_reward=False)
        cb = E.EvalCallback(ev, n_eval_episodes=2, eval_freq=1, verbose=0)
        cb._record(ppo, E.EvalResult(episode_rewards=[1.5, 2.5], episode_lengths=[10, 12]), ppo.num_timesteps, None)
        out["eval_mean_reward"] = float(cb.last_mean_reward)
    return out


def main():
    forced = bool(os.environ.get("FW_DIST_FORCE"))
    world, rank, local = R.init_distributed_from_env()
    import torch.distributed as td
    res = {"forced": forced, "initialised": bool(td.is_initialized()), "backend": td.get_backend() if td.is_initialized() else None, "world": world}
    for mode in ("replicated", "allreduce"):
        res[mode] = run(mode, forced)
    if td.is_initialized():
        td.barrier(); td.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
