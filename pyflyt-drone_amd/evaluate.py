"""Evaluation harness: the reference's acceptance metrics on the device envs.

Mirrors, for the device-resident envs:

* SB3 ``evaluate_policy(model, eval_env, n_eval_episodes, deterministic=True,
  return_episode_rewards=True, callback=...)`` as the reference calls it
  (train/train_Fixedwing_Waypoints_v3.py:161-170): episodes are counted per env with the
  targets ``(n_eval_episodes + i) // n_envs`` so that short episodes of fast envs do not bias
  the sample, rewards are the *un-normalised* env rewards, observations are normalised with
  frozen statistics (``training=False, norm_reward=False`` eval wrapper, ``:264-270``);
* ``sync_envs_normalization`` before every evaluation (``:146-150``);
* the logged scalars of ``WaypointEvalCallback._on_step`` (``:196-214``): ``eval/mean_reward``,
  ``eval/mean_ep_length``, ``eval/wp{i}_reach_rate`` = mean(num_targets_reached >= i),
  ``eval/success_rate`` = mean(is_success); the ObjLock scripts add ``duck_strike_rate``
  (train/train_objlock.py:163-167, eval/eval_objlock.py:260-325);
* ``evaluations.npz`` (timesteps / results / ep_lengths / successes) and ``best_model`` on a
  new best mean reward (``:172-190, 216-222``).

Everything per step stays on the device; the host reads one small ``dones`` mask per
vec-step (the episode bookkeeping is host-side like SB3's).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import config as K


def sync_envs_normalization(train_env, eval_env) -> None:
    """Copy the running statistics of the training wrapper into the eval wrapper
    (SB3 ``sync_envs_normalization``)."""
    eval_env.obs_rms.load_state_dict(train_env.obs_rms.state_dict())
    eval_env.ret_rms.load_state_dict(train_env.ret_rms.state_dict())


@dataclass
class EvalResult:
    episode_rewards: List[float]
    episode_lengths: List[int]
    num_targets_reached: List[int] = field(default_factory=list)
    is_success: List[bool] = field(default_factory=list)
    duck_strike: List[bool] = field(default_factory=list)

    @property
    def mean_reward(self) -> float: return float(np.mean(self.episode_rewards))
    @property
    def std_reward(self) -> float: return float(np.std(self.episode_rewards))
    @property
    def mean_ep_length(self) -> float: return float(np.mean(self.episode_lengths))
    @property
    def std_ep_length(self) -> float: return float(np.std(self.episode_lengths))

    def scalars(self, num_targets_total: int = 0, has_duck: bool = False) -> Dict[str, float]:
        """The ``eval/*`` scalars the reference's callbacks record."""
        out = {"eval/mean_reward": self.mean_reward, "eval/mean_ep_length": self.mean_ep_length}
        if self.num_targets_reached and num_targets_total > 0:
            reached = np.asarray(self.num_targets_reached, dtype=np.float64)
            for i in range(1, num_targets_total + 1):
                out[f"eval/wp{i}_reach_rate"] = float(np.mean(reached >= i))
        if self.is_success:
            out["eval/success_rate"] = float(np.mean(self.is_success))
        if has_duck and self.duck_strike:
            out["eval/duck_strike_rate"] = float(np.mean(self.duck_strike))
        return out


@torch.no_grad()
def evaluate_policy(policy, env, n_eval_episodes: int = 10, deterministic: bool = True,
                    callback: Optional[Callable[[dict], None]] = None, max_vec_steps: Optional[int] = None,
                    generator: Optional[torch.Generator] = None, use_graph: Optional[bool] = None) -> EvalResult:
    """Run ``policy`` on ``env`` (a :class:`~.rollout.VecNormalizeDevice` over a device env,
    normally with ``training=False, norm_reward=False``) until ``n_eval_episodes`` episodes are
    complete.  ``callback(info_dict)`` is called for every finished episode with the keys the
    reference's callbacks read (``num_targets_reached``, ``is_success``, ``duck_strike``, ...).

    Deterministic evaluations on the GPU run as replays of a captured hipGraph of 8 vec-steps with the episode bookkeeping
    on the device (``use_graph``; default: whenever possible): an evaluation of 16 envs flying 1800-step episodes is ~30
    framework ops per step, and read back after every step it cost as much wall clock as 0.4 M training steps.  Both paths
    return the same episodes in the same order; ``callback`` is then called once the evaluation is over."""
    venv = env.venv
    n = env.num_envs
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=np.int64)
    if use_graph is None:
        use_graph = deterministic and generator is None and torch.device(env.device).type == "cuda" and hasattr(venv, "step_tensor")
    if use_graph:
        return _evaluate_replayed(policy, env, targets, callback, max_vec_steps)
    counts = np.zeros(n, dtype=np.int64)
    cur_rew = torch.zeros(n, dtype=torch.float64, device=env.device)
    cur_len = torch.zeros(n, dtype=torch.int64, device=env.device)
    res = EvalResult([], [])
    has_info = hasattr(venv, "info")
    is_objlock = getattr(getattr(venv, "cfg", None), "task", K.FW_TASK_WAYPOINTS) != K.FW_TASK_WAYPOINTS
    obs = env.reset()
    steps = 0
    while (counts < targets).any():
        actions, _, _ = policy(obs, deterministic=deterministic, generator=generator)
        clipped = actions.clamp(-1.0, 1.0).to(venv.torch_dtype)
        obs, _, dones, _, _ = env.step(clipped)
        cur_rew += venv.rewards.to(torch.float64)        # un-normalised reward of the wrapped env
        cur_len += 1
        d = dones.cpu().numpy()
        if d.any():
            idx = np.nonzero(d)[0]
            rew_h, len_h = cur_rew.cpu().numpy(), cur_len.cpu().numpy()
            info_h = venv.info.cpu().numpy() if has_info else None
            for i in idx:
                if counts[i] < targets[i]:
                    counts[i] += 1
                    res.episode_rewards.append(float(rew_h[i])); res.episode_lengths.append(int(len_h[i]))
                    info = {"episode": {"r": float(rew_h[i]), "l": int(len_h[i])}}
                    if info_h is not None:
                        info["num_targets_reached"] = int(info_h[i, K.INFO_NUM_TARGETS_REACHED])
                        info["collision"] = bool(info_h[i, K.INFO_COLLISION])
                        info["out_of_bounds"] = bool(info_h[i, K.INFO_OUT_OF_BOUNDS])
                        info["env_complete"] = bool(info_h[i, K.INFO_ENV_COMPLETE])
                        res.num_targets_reached.append(info["num_targets_reached"])
                        if is_objlock:
                            info["duck_strike"] = bool(info_h[i, K.INFO_DUCK_STRIKE])
                            info["is_success"] = bool(info_h[i, K.INFO_IS_SUCCESS])
                            res.duck_strike.append(info["duck_strike"])
                        else:
                            info["is_success"] = info["env_complete"]
                        res.is_success.append(info["is_success"])
                    if callback is not None:
                        callback(info)
            m = torch.as_tensor(d, device=env.device)
            cur_rew.masked_fill_(m, 0.0); cur_len.masked_fill_(m, 0)
        steps += 1
        if max_vec_steps is not None and steps >= max_vec_steps:
            break
    return res


def _episode_info(res: "EvalResult", rew: float, length: int, info_row, is_objlock: bool) -> dict:
    info = {"episode": {"r": rew, "l": length}}
    if info_row is not None:
        info["num_targets_reached"] = int(info_row[K.INFO_NUM_TARGETS_REACHED])
        info["collision"] = bool(info_row[K.INFO_COLLISION])
        info["out_of_bounds"] = bool(info_row[K.INFO_OUT_OF_BOUNDS])
        info["env_complete"] = bool(info_row[K.INFO_ENV_COMPLETE])
        res.num_targets_reached.append(info["num_targets_reached"])
        if is_objlock:
            info["duck_strike"] = bool(info_row[K.INFO_DUCK_STRIKE])
            info["is_success"] = bool(info_row[K.INFO_IS_SUCCESS])
            res.duck_strike.append(info["duck_strike"])
        else:
            info["is_success"] = info["env_complete"]
        res.is_success.append(info["is_success"])
    return info


_REPLAY_STEPS = 8


def _evaluate_replayed(policy, env, targets: np.ndarray, callback, max_vec_steps) -> "EvalResult":
    """evaluate_policy with the loop body captured: policy forward, env step, normalisation and the per-episode bookkeeping
    (reward / length accumulators, the slot of the episode that just ended, its info row) are device ops on fixed buffers;
    the host looks at the episode counters once per replay of ``_REPLAY_STEPS`` steps."""
    venv, n, dev = env.venv, env.num_envs, env.device
    E = max(int(targets.max()), 1)
    has_info = hasattr(venv, "info")
    is_objlock = getattr(getattr(venv, "cfg", None), "task", K.FW_TASK_WAYPOINTS) != K.FW_TASK_WAYPOINTS
    tg = torch.as_tensor(targets, device=dev)
    ar = torch.arange(n, device=dev)
    obs = env.reset().clone()
    counts = torch.zeros(n, dtype=torch.int64, device=dev)
    cur_rew = torch.zeros(n, dtype=torch.float64, device=dev)
    cur_len = torch.zeros(n, dtype=torch.int64, device=dev)
    step_ctr = torch.zeros((), dtype=torch.int64, device=dev)
    fin_rew = torch.zeros((n, E), dtype=torch.float64, device=dev)
    fin_len = torch.zeros((n, E), dtype=torch.int64, device=dev)
    fin_step = torch.zeros((n, E), dtype=torch.int64, device=dev)
    fin_info = torch.zeros((n, E, venv.info.shape[1]), dtype=venv.info.dtype, device=dev) if has_info else None

    def body():
        actions, _, _ = policy(obs, deterministic=True, generator=None)
        o, _, dones, _, _ = env.step(actions.clamp(-1.0, 1.0).to(venv.torch_dtype))
        obs.copy_(o)
        cur_rew.add_(venv.rewards.to(torch.float64))                 # un-normalised reward of the wrapped env
        cur_len.add_(1); step_ctr.add_(1)
        take = dones & (counts < tg)
        slot = counts.clamp(max=E - 1)
        fin_rew[ar, slot] = torch.where(take, cur_rew, fin_rew[ar, slot])
        fin_len[ar, slot] = torch.where(take, cur_len, fin_len[ar, slot])
        fin_step[ar, slot] = torch.where(take, step_ctr.expand(n), fin_step[ar, slot])
        if has_info:
            fin_info[ar, slot] = torch.where(take[:, None], venv.info, fin_info[ar, slot])
        counts.add_(take.to(torch.int64))
        cur_rew.masked_fill_(dones, 0.0); cur_len.masked_fill_(dones, 0)

    def unfinished() -> bool:
        return bool((counts < tg).any().item())

    steps = 0
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):                                           # warm-up (these are evaluation steps like any other)
            body(); steps += 1
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(_REPLAY_STEPS):
            body()
    steps += _REPLAY_STEPS                                           # (capture does not execute: the first replay does)
    graph.replay()
    while unfinished() and (max_vec_steps is None or steps < max_vec_steps):
        graph.replay(); steps += _REPLAY_STEPS
    torch.cuda.synchronize(dev)
    res = EvalResult([], [])
    c_h = torch.minimum(counts, tg).cpu().numpy()
    rew_h, len_h, step_h = fin_rew.cpu().numpy(), fin_len.cpu().numpy(), fin_step.cpu().numpy()
    info_h = fin_info.cpu().numpy() if has_info else None
    order = sorted((int(step_h[i, k]), i, k) for i in range(n) for k in range(int(c_h[i])))        # as they ended: by step, then env
    for _, i, k in order:
        res.episode_rewards.append(float(rew_h[i, k])); res.episode_lengths.append(int(len_h[i, k]))
        info = _episode_info(res, float(rew_h[i, k]), int(len_h[i, k]), info_h[i, k] if has_info else None, is_objlock)
        if callback is not None:
            callback(info)
    return res


class EvalCallback:
    """``WaypointEvalCallback`` / SB3 ``EvalCallback`` for :meth:`rollout.PPO.learn`: every
    ``eval_freq`` vec-steps of training (the reference passes ``10000 // num_envs``) sync the
    normaliser, evaluate, append to ``evaluations.npz``, keep ``best_model``."""

    def __init__(self, eval_env, n_eval_episodes: int = 5, eval_freq: int = 10000, log_path: Optional[str] = None,
                 best_model_save_path: Optional[str] = None, deterministic: bool = True, num_targets_total: int = 0,
                 verbose: int = 0):
        self.eval_env, self.n_eval_episodes, self.eval_freq = eval_env, n_eval_episodes, max(int(eval_freq), 1)
        self.log_path = os.path.join(log_path, "evaluations") if log_path else None
        self.best_model_save_path, self.deterministic = best_model_save_path, deterministic
        self.num_targets_total, self.verbose = num_targets_total, verbose
        self.best_mean_reward, self.last_mean_reward = -np.inf, -np.inf
        self.evaluations_timesteps: List[int] = []
        self.evaluations_results: List[List[float]] = []
        self.evaluations_length: List[List[int]] = []
        self.evaluations_successes: List[List[bool]] = []
        self.last_scalars: Dict[str, float] = {}
        self._next_eval_calls = self.eval_freq
        self.n_evals = 0

    def on_rollout_end(self, ppo) -> bool:
        n_calls = ppo.num_timesteps // max(ppo.env.num_envs * ppo.world_size, 1)      # vec-steps so far (SB3 n_calls)
        if n_calls < self._next_eval_calls:
            return True
        while self._next_eval_calls <= n_calls:
            self._next_eval_calls += self.eval_freq
        from . import checkpoint
        sync_envs_normalization(ppo.env, self.eval_env)
        r = evaluate_policy(ppo.policy, self.eval_env, self.n_eval_episodes, deterministic=self.deterministic)
        self.n_evals += 1
        self.evaluations_timesteps.append(ppo.num_timesteps)
        self.evaluations_results.append(r.episode_rewards); self.evaluations_length.append(r.episode_lengths)
        writer = ppo.rank == 0               # multi-process job: every rank evaluates (identical weights), ONE rank writes files
        if ppo.world_size > 1:               # ... and every rank keeps rank 0's figure, so best_mean_reward agrees everywhere
            import torch.distributed as td
            t = torch.tensor([r.mean_reward], dtype=torch.float64, device=ppo.device if td.get_backend() == "nccl" else "cpu")
            td.broadcast(t, src=0)
            r.mean_reward_override = float(t.item())
        if self.log_path is not None and writer:
            os.makedirs(os.path.dirname(self.log_path), exist_ok=True)
            kw = {}
            if r.is_success:
                self.evaluations_successes.append(r.is_success); kw = dict(successes=np.array(self.evaluations_successes, dtype=object))
            np.savez(self.log_path, timesteps=self.evaluations_timesteps, results=np.array(self.evaluations_results, dtype=object),
                     ep_lengths=np.array(self.evaluations_length, dtype=object), **kw)
        mean_reward = getattr(r, "mean_reward_override", r.mean_reward)
        self.last_mean_reward = mean_reward
        is_objlock = getattr(getattr(self.eval_env.venv, "cfg", None), "task", 0) != K.FW_TASK_WAYPOINTS
        self.last_scalars = r.scalars(self.num_targets_total, has_duck=is_objlock)
        self.last_scalars["time/total_timesteps"] = ppo.num_timesteps
        if self.verbose and writer:
            print(f"Eval num_timesteps={ppo.num_timesteps}, episode_reward={r.mean_reward:.2f} +/- {r.std_reward:.2f}")
            print(f"Episode length: {r.mean_ep_length:.2f} +/- {r.std_ep_length:.2f}")
        if mean_reward > self.best_mean_reward:
            self.best_mean_reward = mean_reward
            if self.best_model_save_path is not None and writer:
                checkpoint.save(os.path.join(self.best_model_save_path, "best_model.pt"), ppo, include_env_state=False)
        return True
