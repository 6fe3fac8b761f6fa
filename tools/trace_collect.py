"""Dev tool: where one fw_collect_step launch spends its time (wall-clock stamps written by every workgroup, 10 ns ticks).
    python tools/trace_collect.py [waypoints|objlock|combined] [envs]
Prints, relative to the first stamp of the launch: when the act waves have their weights / inputs / forward pass / published,
when the step waves start, how long they wait for their actions, when their step is over, and how long the statistics tail
(partials -> group fold -> final merge) takes."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K, rollout as R

task = sys.argv[1] if len(sys.argv) > 1 else "waypoints"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = {"waypoints": K.train_waypoints_v3_config, "objlock": K.train_objlock_config, "combined": K.train_waypoint_objlock_config}[task]()
env = P.FixedwingVecEnv(cfg, n, seed=42)
ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=8, batch_size=64, n_epochs=1, use_graphs=False, one_launch_collect=True))
assert ppo._one_launch
for _ in range(3):
    ppo.collect_rollouts()
n_chunks = (n + 15) // 16
n_act = (2 * n_chunks + 1 + 7) & ~7
nblk = (n + 63) // 64 * 64 // 8
pw = 2 * int(env.observation_space.shape[0]) + 2
grid = n_act + 2 * nblk + pw
ppo._trace = torch.zeros((grid, 8), dtype=torch.int64, device=env.device)
rows = []
for rep in range(5):
    ppo._trace.zero_()
    ppo.collect_rollouts()                     # 8 launches: the stamps of the last one remain
    torch.cuda.synchronize()
    t = ppo._trace.cpu().numpy().astype(np.float64) * 0.01       # us (100 MHz)
    t0 = t[:, 0][t[:, 0] > 0].min()
    act, step = t[:2 * n_chunks], t[n_act:n_act + nblk]
    pol, val = act[0::2], act[1::2]
    rel = lambda x: x - t0
    rows.append(dict(
        act_start_max=rel(act[:, 0]).max(), act_stats_known=np.mean(act[:, 1] - act[:, 0]), act_column_consts=np.mean(act[:, 6] - act[:, 1]), act_normalised=np.mean(act[:, 5] - act[:, 6]), act_tile_in_lds=np.mean(act[:, 2] - act[:, 5]),
        act_forward=np.mean(act[:, 3] - act[:, 2]), pol_publish_mean=rel(pol[:, 4]).mean(), pol_publish_max=rel(pol[:, 4]).max(),
        val_inputs_max=rel(val[:, 2]).max(), val_done_max=rel(val[:, 4]).max(),
        step_start_mean=rel(step[:, 0]).mean(), step_start_max=rel(step[:, 0]).max(), wait_begin_mean=rel(step[:, 1]).mean(),
        wait_end_mean=rel(step[:, 2]).mean(), wait_end_max=rel(step[:, 2]).max(), wait_us_mean=np.mean(step[:, 2] - step[:, 1]),
        step_body_mean=np.mean(step[:, 3] - step[:, 2]), step_over_max=rel(step[:, 3]).max(), partials_max=rel(step[:, 4]).max(),
        fold_start_max=rel(t[n_act + 2 * nblk:, 0]).max(), launch_end=rel(t[n_act:, 7].max())))
keys = rows[0].keys()
print(f"{task} {n} envs: grid {grid} = {n_act} act + {nblk} step + {nblk} worker + {pw} fold workgroups; us from the launch's first stamp, median of 5 launches")
for k in keys:
    print(f"  {k:18s} {np.median([r[k] for r in rows]):8.2f}")
