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
from .rollout import policy_inputs


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
                    generator: Optional[torch.Generator] = None, use_graph: Optional[bool] = None,
                    use_fused: Optional[bool] = None) -> EvalResult:
    """Run ``policy`` on ``env`` (a :class:`~.rollout.VecNormalizeDevice` over a device env,
    normally with ``training=False, norm_reward=False``) until ``n_eval_episodes`` episodes are
    complete.  ``callback(info_dict)`` is called for every finished episode with the keys the
    reference's callbacks read (``num_targets_reached``, ``is_success``, ``duck_strike``, ...).

    Deterministic evaluations on the GPU run as replays of a captured hipGraph of 8 vec-steps with the episode bookkeeping
    on the device (``use_graph``; default: whenever possible): an evaluation of 16 envs flying 1800-step episodes is ~30
    framework ops per step, and read back after every step it cost as much wall clock as 0.4 M training steps.  Both paths
    return the same episodes in the same order; ``callback`` is then called once the evaluation is over.

    ``use_fused`` (default: whenever possible -- the reference's MlpPolicy on a device env with the 8-lane mapping, an
    evaluation normaliser with frozen statistics): a vec-step of the replayed evaluation is ONE ``fw_collect_step`` launch in
    its deterministic, statistics-frozen form (normalisation, policy forward on the matrix cores, clip, env step) instead of
    ~20 framework ops; the policy's actions then agree with the torch forward to fp32 rounding, not to the bit."""
    venv = env.venv
    n = env.num_envs
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=np.int64)
    if use_graph is None:
        use_graph = (deterministic and generator is None and torch.device(env.device).type == "cuda" and hasattr(venv, "step_tensor")
                     and not getattr(policy, "uses_image", False))        # (a CNN front end keeps MIOpen out of captures)
    if use_graph:
        return ReplayedEvaluation(policy, env, targets, callback, use_fused=use_fused).run(max_vec_steps)
    counts = np.zeros(n, dtype=np.int64)
    cur_rew = torch.zeros(n, dtype=torch.float64, device=env.device)
    cur_len = torch.zeros(n, dtype=torch.int64, device=env.device)
    res = EvalResult([], [])
    has_info = hasattr(venv, "info")
    is_objlock = getattr(getattr(venv, "cfg", None), "task", K.FW_TASK_WAYPOINTS) != K.FW_TASK_WAYPOINTS
    obs = env.reset()
    steps = 0
    while (counts < targets).any():
        actions, _, _ = policy(obs, deterministic=deterministic, generator=generator, **policy_inputs(policy, env))
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


class ReplayedEvaluation:
    """evaluate_policy with the loop body captured: policy forward, env step, normalisation and the per-episode bookkeeping
    (reward / length accumulators, the slot of the episode that just ended, its info row) are device ops on fixed buffers.

    ``run()`` drives it from the host (one look at the episode counters per replay of ``_REPLAY_STEPS`` steps).
    ``launch()`` enqueues the WHOLE evaluation on a side stream -- as many replays as the longest possible episodes need,
    ``episodes per env x (max_steps + 2)`` vec-steps -- and returns at once: the evaluation (a few envs) then runs beside the
    training that continues on the main stream (the PPO update keeps eight of the 256 CUs busy); ``ready()`` / ``result()``
    collect it.  The policy handed in must not change while it runs (EvalCallback evaluates a copy of the weights)."""

    def __init__(self, policy, env, targets: np.ndarray, callback=None, use_fused: Optional[bool] = None):
        self.policy, self.env, self.callback = policy, env, callback
        venv, n, dev = env.venv, env.num_envs, env.device
        self.fused = self._fused_applies(policy, env) if use_fused is None else bool(use_fused)
        if self.fused and not self._fused_applies(policy, env):
            raise ValueError("use_fused=True needs the MlpPolicy, a device env on the 8-lane mapping and an evaluation normaliser (training=False)")
        self.venv, self.n, self.dev, self.targets = venv, n, dev, targets
        self.E = E = max(int(targets.max()), 1)
        self.has_info = hasattr(venv, "info")
        self.is_objlock = getattr(getattr(venv, "cfg", None), "task", K.FW_TASK_WAYPOINTS) != K.FW_TASK_WAYPOINTS
        self.tg = torch.as_tensor(targets, device=dev)
        self.ar = torch.arange(n, device=dev)
        self.counts = torch.zeros(n, dtype=torch.int64, device=dev)
        self.cur_rew = torch.zeros(n, dtype=torch.float64, device=dev)
        self.cur_len = torch.zeros(n, dtype=torch.int64, device=dev)
        self.step_ctr = torch.zeros((), dtype=torch.int64, device=dev)
        self.fin_rew = torch.zeros((n, E), dtype=torch.float64, device=dev)
        self.fin_len = torch.zeros((n, E), dtype=torch.int64, device=dev)
        self.fin_step = torch.zeros((n, E), dtype=torch.int64, device=dev)
        self.fin_info = torch.zeros((n, E, venv.info.shape[1]), dtype=venv.info.dtype, device=dev) if self.has_info else None
        self.obs = None
        self.side = torch.cuda.Stream(device=dev)
        self.done_event = None
        self.steps = 0
        if self.fused:
            self._fused_setup()

    # ---- a vec-step as ONE fw_collect_step launch (deterministic, statistics frozen) ----
    @staticmethod
    def _fused_applies(policy, env) -> bool:
        from . import _lib
        from .rollout import FusedPpoUpdate
        venv = env.venv
        if not (hasattr(venv, "_h") and hasattr(venv, "step_tensor") and torch.device(env.device).type == "cuda"):
            return False
        if env.training or not env.norm_obs or not FusedPpoUpdate.fits(policy, env.obs_dim, torch.device(env.device)):
            return False
        if env.obs_dim > 62:                   # fw_collect_step's act waves take up to 62 features (fits() allows the update's 64): torch replay path
            return False
        return int(_lib.lib().fw_lanes_per_env(venv._h)) in (8, 16)

    def _fused_setup(self) -> None:
        from . import _lib
        from .rollout import FusedPpoUpdate
        env, venv, dev, n = self.env, self.venv, self.dev, self.n
        L = _lib.lib()
        f = FusedPpoUpdate(self.policy, None, env.obs_dim)
        f.load_params_from_torch()
        self._flat = f.flat                                         # the kernel's parameter image of the policy being evaluated
        self._act_env = torch.full((n, 4), float("nan"), dtype=venv.torch_dtype, device=dev)      # NaN = "not there yet" (fw_collect_step)
        self._act_raw = torch.zeros((n, 4), dtype=torch.float32, device=dev)                     # rollout-buffer rows the launch fills: not looked at
        self._logp, self._val = torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)
        self._rng = torch.zeros(2, dtype=torch.int64, device=dev)
        nb = int(L.fw_collect_step_workspace_bytes(venv._h))
        self._ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
        self._ws_ready = False

    def _fused_step(self) -> None:
        import ctypes as C
        from . import _lib
        from .rollout import _stream
        env, venv = self.env, self.venv
        L, st = _lib.lib(), _stream(self.dev)
        if not self._ws_ready:
            _lib.check(L.fw_collect_workspace_init(venv._h, self._ws.data_ptr(), self._ws.numel() * 8, st), venv._h)
            self._ws_ready = True
        a = K.FwCollectArgs()
        a.params = self._flat.data_ptr()
        a.obs_mean, a.obs_var, a.obs_count = env.obs_rms.mean.data_ptr(), env.obs_rms.var.data_ptr(), env.obs_rms.count.data_ptr()
        a.returns = env.returns.data_ptr()
        a.ret_mean, a.ret_var, a.ret_count = env.ret_rms.mean.data_ptr(), env.ret_rms.var.data_ptr(), env.ret_rms.count.data_ptr()
        a.rng = self._rng.data_ptr()
        a.act_raw, a.logp, a.value = self._act_raw.data_ptr(), self._logp.data_ptr(), self._val.data_ptr()
        a.act_env = self._act_env.data_ptr()
        a.obs, a.reward = venv.obs.data_ptr(), venv.rewards.data_ptr()
        a.terminated, a.truncated = venv.terminated.data_ptr(), venv.truncated.data_ptr()
        a.terminal_obs, a.info_i32 = venv.terminal_obs.data_ptr(), venv.info.data_ptr()
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel() * 8
        a.gamma = float(env.gamma)
        a.clip_obs, a.eps_obs, a.clip_reward, a.eps_reward = float(env.clip_obs), float(env.epsilon), float(env.clip_reward), float(env.epsilon)
        a.update_obs, a.update_ret, a.norm_reward, a.deterministic = 0, 0, 0, 1
        _lib.check(L.fw_collect_step(venv._h, C.byref(a), st), venv._h)
        # ... and the episode bookkeeping of the step in one more (fw_eval_track)
        fi = self.fin_info
        _lib.check(L.fw_eval_track(venv.rewards.data_ptr(), int(venv.rewards.dtype == torch.float64), venv.terminated.data_ptr(),
                                   venv.truncated.data_ptr(), venv.info.data_ptr() if fi is not None else None,
                                   int(venv.info.shape[1]) if fi is not None else 0, self.tg.data_ptr(), self.counts.data_ptr(),
                                   self.cur_rew.data_ptr(), self.cur_len.data_ptr(), self.step_ctr.data_ptr(), self.fin_rew.data_ptr(),
                                   self.fin_len.data_ptr(), self.fin_step.data_ptr(), fi.data_ptr() if fi is not None else None,
                                   self.n, self.E, st))

    def _fused_check(self) -> None:
        """fw_collect_step's status word: a wait inside one of the launches ran out -> the evaluation is void ("returns or raises")"""
        import ctypes as C
        from . import _lib
        from .rollout import _stream
        if not self.fused or not self._ws_ready:
            return
        stw = C.c_uint32(0)
        _lib.check(_lib.lib().fw_collect_status(self.venv._h, self._ws.data_ptr(), self._ws.numel() * 8, C.byref(stw), _stream(self.dev)), self.venv._h)
        if stw.value:
            raise RuntimeError(f"fw_collect_step: status word {stw.value} during the evaluation; its figures are void")

    def _body(self):
        venv, ar, tg, E, counts = self.venv, self.ar, self.tg, self.E, self.counts
        if self.fused:
            self._fused_step()
            return
        else:
            actions, _, _ = self.policy(self.obs, deterministic=True, generator=None, **policy_inputs(self.policy, self.env))
            o, _, dones, _, _ = self.env.step(actions.clamp(-1.0, 1.0).to(venv.torch_dtype))
            self.obs.copy_(o)
        self.cur_rew.add_(venv.rewards.to(torch.float64))            # un-normalised reward of the wrapped env
        self.cur_len.add_(1); self.step_ctr.add_(1)
        take = dones & (counts < tg)
        slot = counts.clamp(max=E - 1)
        self.fin_rew[ar, slot] = torch.where(take, self.cur_rew, self.fin_rew[ar, slot])
        self.fin_len[ar, slot] = torch.where(take, self.cur_len, self.fin_len[ar, slot])
        self.fin_step[ar, slot] = torch.where(take, self.step_ctr.expand(self.n), self.fin_step[ar, slot])
        if self.has_info:
            self.fin_info[ar, slot] = torch.where(take[:, None], venv.info, self.fin_info[ar, slot])
        counts.add_(take.to(torch.int64))
        self.cur_rew.masked_fill_(dones, 0.0); self.cur_len.masked_fill_(dones, 0)

    def _begin(self):
        """reset, two eager steps (they are evaluation steps like any other), capture: on the side stream"""
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            self.obs = self.env.reset().clone()
            for _ in range(2):
                self._body(); self.steps += 1
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.side):
                for _ in range(_REPLAY_STEPS):
                    self._body()

    def run(self, max_vec_steps: Optional[int] = None) -> "EvalResult":
        self._begin()
        with torch.cuda.stream(self.side):
            while True:
                self.graph.replay(); self.steps += _REPLAY_STEPS
                if not bool((self.counts < self.tg).any().item()) or (max_vec_steps is not None and self.steps >= max_vec_steps):
                    break
        torch.cuda.current_stream(self.dev).wait_stream(self.side)
        return self.result()

    def launch(self, bound_vec_steps: int) -> "ReplayedEvaluation":
        self._begin()
        with torch.cuda.stream(self.side):
            for _ in range(-(-max(bound_vec_steps - 2, 1) // _REPLAY_STEPS)):
                self.graph.replay(); self.steps += _REPLAY_STEPS
            self.done_event = torch.cuda.Event()
            self.done_event.record(self.side)
        return self

    def ready(self) -> bool:
        return self.done_event is None or self.done_event.query()

    def result(self) -> "EvalResult":
        if self.done_event is not None:
            self.done_event.synchronize()
        else:
            self.side.synchronize()
        self._fused_check()
        n, has_info = self.n, self.has_info
        res = EvalResult([], [])
        c_h = torch.minimum(self.counts, self.tg).cpu().numpy()
        rew_h, len_h, step_h = self.fin_rew.cpu().numpy(), self.fin_len.cpu().numpy(), self.fin_step.cpu().numpy()
        info_h = self.fin_info.cpu().numpy() if has_info else None
        order = sorted((int(step_h[i, k]), i, k) for i in range(n) for k in range(int(c_h[i])))    # as they ended: by step, then env
        for _, i, k in order:
            res.episode_rewards.append(float(rew_h[i, k])); res.episode_lengths.append(int(len_h[i, k]))
            info = _episode_info(res, float(rew_h[i, k]), int(len_h[i, k]), info_h[i, k] if has_info else None, self.is_objlock)
            if self.callback is not None:
                self.callback(info)
        return res


def start_evaluation(policy, env, n_eval_episodes: int = 10, callback=None) -> ReplayedEvaluation:
    """Asynchronous :func:`evaluate_policy` (deterministic, device envs whose config bounds the episode length): returns a
    running :class:`ReplayedEvaluation`; ``.result()`` waits for it."""
    n = env.num_envs
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=np.int64)
    bound = int(targets.max()) * (K.max_steps(env.venv.cfg) + 2)
    return ReplayedEvaluation(policy, env, targets, callback).launch(bound)


class EvalCallback:
    """``WaypointEvalCallback`` / SB3 ``EvalCallback`` for :meth:`rollout.PPO.learn`: every
    ``eval_freq`` vec-steps of training (the reference passes ``10000 // num_envs``) sync the
    normaliser, evaluate, append to ``evaluations.npz``, keep ``best_model``.

    ``overlap=True`` (single-process jobs on the GPU; off by default): the evaluation is *launched* -- a copy of the
    weights, the normaliser statistics, then :func:`start_evaluation` on a side stream -- and the training goes on; its
    figures are logged (under the timestep count at which it was launched) when a later callback finds it finished, at the
    latest before the next evaluation starts and when training ends.  ``best_model`` is written from the snapshot taken at
    launch, so it holds the evaluated weights, not the ones trained since.  Same figures as the synchronous form
    (tests/test_eval_checkpoint_gpu.py) -- but measured SLOWER on the training examples (combined 208 k -> 186 k,
    waypoints 209 k -> 202 k env-steps/s over whole runs): without a host in the loop every env is stepped for the longest
    possible episodes, and those ~1800 small launches on a second queue cost the training stream more than the 0.1 s of
    waiting they replace.  The default is therefore the synchronous evaluation, replayed as hipGraphs of 8 vec-steps."""

    def __init__(self, eval_env, n_eval_episodes: int = 5, eval_freq: int = 10000, log_path: Optional[str] = None,
                 best_model_save_path: Optional[str] = None, deterministic: bool = True, num_targets_total: int = 0,
                 verbose: int = 0, overlap: Optional[bool] = None):
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
        self.overlap = overlap
        self._pending = None              # (job, num_timesteps at launch, checkpoint snapshot or None)
        self._policy_copy = None
        self._seed0, self._n_launched = None, 0

    def _can_overlap(self, ppo) -> bool:
        if not self.overlap:
            return False
        venv = self.eval_env.venv
        return (self.deterministic and ppo.world_size == 1 and torch.device(self.eval_env.device).type == "cuda"
                and hasattr(venv, "step_tensor") and hasattr(venv, "cfg"))

    def on_rollout_end(self, ppo) -> bool:
        if self._pending is not None and self._pending[0].ready():
            self._finish(ppo)
        n_calls = ppo.num_timesteps // max(ppo.env.num_envs * ppo.world_size, 1)      # vec-steps so far (SB3 n_calls)
        if n_calls < self._next_eval_calls:
            return True
        while self._next_eval_calls <= n_calls:
            self._next_eval_calls += self.eval_freq
        from . import checkpoint
        if self._pending is not None:
            self._finish(ppo)                                    # one evaluation at a time: its env and weight copy are reused
        sync_envs_normalization(ppo.env, self.eval_env)
        # the k-th evaluation draws its scenarios from (seed of the eval env + k): its episodes then do not depend on how many
        # steps the earlier evaluations happened to run past their last episode (the overlapped form runs every env for the
        # longest possible episodes), and two runs of the same training evaluate on the same scenarios
        venv = self.eval_env.venv
        if hasattr(venv, "seed") and hasattr(venv, "seed_value"):
            if self._seed0 is None:
                self._seed0 = int(venv.seed_value)
            venv.seed(self._seed0 + self._n_launched)
        self._n_launched += 1
        writer = ppo.rank == 0
        snap = checkpoint.snapshot(ppo, include_env_state=False) if (self.best_model_save_path is not None and writer) else None
        if self._can_overlap(ppo):
            import copy
            if self._policy_copy is None:
                self._policy_copy = copy.deepcopy(ppo.policy)
            else:
                self._policy_copy.load_state_dict(ppo.policy.state_dict())
            job = start_evaluation(self._policy_copy, self.eval_env, self.n_eval_episodes)
            self._pending = (job, ppo.num_timesteps, snap)
        else:
            r = evaluate_policy(ppo.policy, self.eval_env, self.n_eval_episodes, deterministic=self.deterministic)
            self._record(ppo, r, ppo.num_timesteps, snap)
        return True

    def on_training_end(self, ppo) -> None:
        if self._pending is not None:
            self._finish(ppo)

    def _finish(self, ppo) -> None:
        job, timesteps, snap = self._pending
        self._pending = None
        self._record(ppo, job.result(), timesteps, snap)

    def _record(self, ppo, r: EvalResult, timesteps: int, snap) -> None:
        from . import checkpoint
        self.n_evals += 1
        self.evaluations_timesteps.append(timesteps)
        self.evaluations_results.append(r.episode_rewards); self.evaluations_length.append(r.episode_lengths)
        writer = ppo.rank == 0               # multi-process job: every rank evaluates (identical weights), ONE rank writes files
        from .rollout import _dist
        if _dist() is not None:              # ... and every rank keeps rank 0's figure, so best_mean_reward agrees everywhere
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
        self.last_scalars["time/total_timesteps"] = timesteps
        if self.verbose and writer:
            print(f"Eval num_timesteps={timesteps}, episode_reward={r.mean_reward:.2f} +/- {r.std_reward:.2f}")
            print(f"Episode length: {r.mean_ep_length:.2f} +/- {r.std_ep_length:.2f}")
        if mean_reward > self.best_mean_reward:
            self.best_mean_reward = mean_reward
            if snap is not None:
                checkpoint.write(snap, os.path.join(self.best_model_save_path, "best_model.pt"))
