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
# where the slowest act waves lose their time: phase means of the policy waves by decile of their publishing time (last launch)
pol = t[:2 * n_chunks][0::2]
order = np.argsort(pol[:, 4])
k = max(1, len(order) // 10)
for name, idx in (("fastest 10 %", order[:k]), ("median 10 %", order[len(order) // 2 - k // 2:len(order) // 2 + k // 2 + 1]), ("slowest 10 %", order[-k:])):
    w = pol[idx]
    print(f"  policy waves, {name}: start {np.mean(w[:, 0] - t0):5.2f}  statistics {np.mean(w[:, 1] - w[:, 0]):5.2f}  tile {np.mean(w[:, 2] - w[:, 1]):5.2f}  "
          f"weights arrive +{np.mean(w[:, 7] - w[:, 2]):5.2f}  forward {np.mean(w[:, 3] - w[:, 7]):5.2f}  sample+publish {np.mean(w[:, 4] - w[:, 3]):5.2f}  published {np.mean(w[:, 4] - t0):5.2f}  "
          f"XCDs {np.bincount((2 * idx) % 8, minlength=8).tolist()}")
act = t[:2 * n_chunks]
xcd = np.arange(2 * n_chunks) % 8
print("  per XCD (act waves; even XCDs carry policy waves, odd ones value waves): us from a wave's start to its statistics / from there to its operands")
for x in range(8):
    w = act[xcd == x]
    print(f"    XCD {x}: statistics {np.mean(w[:, 1] - w[:, 0]):5.2f}  operands {np.mean(w[:, 7] - w[:, 1]):5.2f}  forward {np.mean(w[:, 3] - w[:, 7]):5.2f}  done {np.mean(w[:, 4] - t0):5.2f}")
step = t[n_act:n_act + nblk]
sx = np.arange(nblk) % 8
print("  per XCD (step waves): state loaded after / step body")
for x in range(8):
    w = step[sx == x]
    print(f"    XCD {x}: load {np.mean(w[:, 1] - w[:, 0]):5.2f}  body {np.mean(w[:, 3] - w[:, 2]):5.2f}")

