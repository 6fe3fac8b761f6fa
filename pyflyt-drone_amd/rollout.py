"""Device-resident PPO rollout collector -- the caller side of the env step.

Mirrors what the reference gets from Stable-Baselines3 in
train/train_Fixedwing_Waypoints_v3.py:251-337 (SB3 itself is not a dependency and is
not shipped): ``VecNormalize(norm_obs, norm_reward, clip_obs=10)`` (:260),
``PPO("MlpPolicy", n_steps, batch_size, n_epochs, gamma, gae_lambda, clip_range,
ent_coef, vf_coef, max_grad_norm)`` (:293-310) and its collect/GAE/update cycle.
Everything lives on the GPU: the env step (fw_step), the normaliser (fw_normalize_obs),
the policy MLP (torch-ROCm), the rollout buffer and the GAE scan (fw_gae); nothing makes
a host round trip inside the rollout loop.

Multi-GPU (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI): envs are sharded by rank
(``global_env_offset = rank * N``: scenario, noise and action-sampling streams are keyed on the global env id), the
fused kernels run on every rank exactly as on one GPU, and collectives happen only BETWEEN rollouts --
  * one all-gather of the rank's rollout shard (obs, actions, log-probs, advantages, returns: ~9 MB per 65 536 samples)
    at update time; every rank then runs the SAME minibatch sequence (same permutation seed) on the gathered buffer --
    identical weights by construction, no collective per minibatch (north_star's "all-gather of advantages at update time");
  * one all-reduce per rollout of the normaliser's accumulated batch sums (observations and discounted returns), from
    which every rank re-derives the statistics of ALL envs (``stats_sync="rollout"``); the torch restatement can also
    all-reduce them at every vec-step (``"step"``: the statistics of one big VecNormalize, used by the checkers).
The old data-parallel form (gradient all-reduce per minibatch) is kept as ``PPOConfig.dist_update = "allreduce"``.
"""
from __future__ import annotations

import ctypes as C
import math
import time
import warnings
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _rehearse_one_rank() -> bool:
    """``FW_DIST_FORCE=1``: treat a ONE-rank process group as a sharded job, so that every collective of the multi-GPU path runs
    -- on device tensors, over RCCL -- on a box with a single GPU (RCCL refuses two ranks on one device; gloo rehearsals with two
    ranks take the host-copy branches instead).  Results are those of the single-process job; only the code path differs."""
    import os
    return bool(os.environ.get("FW_DIST_FORCE"))


def _dist():
    import torch.distributed as td
    return td if (td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or _rehearse_one_rank())) else None


def init_distributed_from_env():
    """One process per GPU, as ``torch.distributed.run`` starts them: read WORLD_SIZE / RANK / LOCAL_RANK, bind this process
    to its GPU and join the process group -- backend "nccl" (= RCCL over xGMI on ROCm) unless ``FW_DIST_BACKEND`` says
    otherwise (gloo rehearsals on one device).  Returns ``(world, rank, local_rank)``; ``(1, 0, 0)`` and no process group in a
    plain single-process run."""
    import os
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not (_rehearse_one_rank() and "MASTER_PORT" in os.environ):
        return 1, 0, 0
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FW_DIST_SINGLE_DEVICE"):
        local = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not td.is_initialized():
        backend = os.environ.get("FW_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend)
    return world, rank, local


def n_steps_for(samples_per_update: int, num_envs_per_rank: int, world: int = 1) -> int:
    """Rollout length that HOLDS the reference's samples per update when the envs multiply: ``samples / (envs per rank x
    world)`` (at least 1).  The reference collects 32 x 2048 = 65 536 samples per update (train/train_Fixedwing_Waypoints_v3.py:
    29,35); 4096 envs on one GPU make that 16 steps, 8 GPUs x 4096 envs 2 steps.  It matters twice in a sharded job: the update
    is replicated on every rank (its time is set by the number of sequential minibatches = samples / batch_size x epochs, so
    holding the samples holds the update time while the collection time falls with 1 / world), and PPO's sample efficiency
    is that of the reference's batch."""
    return max(int(samples_per_update) // max(int(num_envs_per_rank) * max(int(world), 1), 1), 1)


def _staged(td, t: torch.Tensor) -> bool:
    """RCCL ("nccl") moves device tensors itself; any other backend (gloo rehearsals) gets a host copy."""
    return t.is_cuda and td.get_backend() != "nccl"


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (no-op in a single-process job)."""
    td = _dist()
    if td is None:
        return t
    if _staged(td, t):
        h = t.detach().cpu()
        td.all_reduce(h)
        t.copy_(h)
    else:
        td.all_reduce(t)
    return t


def all_gather_cat(t: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """Concatenation over ranks (rank order) of ``t`` along ``dim``; one collective.  Single-process: ``t`` itself."""
    td = _dist()
    if td is None:
        return t
    w = td.get_world_size()
    src = t.detach().contiguous()
    if td.get_backend() == "nccl" and src.is_cuda:
        # one direct all-gather on device tensors (RCCL: 7 peers on 7 xGMI links), not a ring of sends
        out = torch.empty((w * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        td.all_gather_into_tensor(out, src)
        out = out.reshape((w,) + tuple(src.shape))
    else:
        h = src.cpu() if src.is_cuda else src                       # gloo rehearsal: host copies
        parts = [torch.empty_like(h) for _ in range(w)]
        td.all_gather(parts, h)
        out = torch.stack(parts).to(t.device)
    return torch.cat(list(out.unbind(0)), dim=dim)


# ---------------------------------------------------------------------------------------------
# running statistics (SB3 common/running_mean_std.py): mean 0, var 1, count = epsilon = 1e-4
# ---------------------------------------------------------------------------------------------
class RunningMeanStd:
    def __init__(self, shape, device, epsilon: float = 1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float64, device=device)
        self.var = torch.ones(shape, dtype=torch.float64, device=device)
        self.count = torch.full((1,), epsilon, dtype=torch.float64, device=device)

    def update(self, x: torch.Tensor) -> None:
        """Chan et al. merge of the batch moments of x[B, ...]; in a multi-GPU job the batch is
        the union of every rank's rows (one all-reduce of [sum, sum of squares, rows])."""
        x = x.to(torch.float64).reshape(x.shape[0], -1)
        s, s2 = x.sum(0), (x * x).sum(0)
        n = torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)   # fill kernel: graph-capturable
        if _dist() is not None:
            buf = all_reduce_sum_(torch.cat([s, s2, n]))
            d = s.numel()
            s, s2, n = buf[:d], buf[d:2 * d], buf[2 * d:]
        self.update_from_sums(s, s2, n)

    def update_from_sums(self, s, s2, n) -> None:
        """Merge a batch given as (column sums, sums of squares, rows) -- the form the kernels accumulate and ranks exchange."""
        s, s2 = s.reshape(-1), s2.reshape(-1)
        bm = s / n
        bv = (s2 / n - bm * bm).clamp_min(0.0)
        self.update_from_moments(bm.reshape(self.mean.shape), bv.reshape(self.var.shape), n)

    def update_from_moments(self, bm, bv, bn) -> None:
        bn, cnt = bn.reshape(-1)[0], self.count[0]          # 0-dim, so scalar statistics keep shape ()
        delta = bm - self.mean
        tot = cnt + bn
        new_mean = self.mean + delta * bn / tot
        m2 = self.var * cnt + bv * bn + delta * delta * cnt * bn / tot
        # in place: a captured rollout graph holds these addresses and must see the accumulated statistics
        self.mean.copy_(new_mean); self.var.copy_(m2 / tot); self.count.copy_(tot.reshape(1))

    def state_dict(self):
        return {"mean": self.mean.cpu(), "var": self.var.cpu(), "count": self.count.cpu()}

    def load_state_dict(self, sd):
        self.mean.copy_(sd["mean"]); self.var.copy_(sd["var"]); self.count.copy_(sd["count"])


class VecNormalizeDevice:
    """SB3 ``VecNormalize`` on device tensors (train/train_Fixedwing_Waypoints_v3.py:260).

    ``step(actions)`` returns ``(obs_f32, reward_f32, dones, terminal_obs_f32)`` all normalised
    exactly like ``VecNormalize.step_wait``: statistics are updated *before* normalising, the
    reward is divided by the std of the discounted return, terminal observations are
    normalised with the same statistics, and ``returns[dones] = 0``."""

    def __init__(self, venv, training: bool = True, norm_obs: bool = True, norm_reward: bool = True,
                 clip_obs: float = 10.0, clip_reward: float = 10.0, gamma: float = 0.99, epsilon: float = 1e-8,
                 use_fused_kernel: Optional[bool] = None, stats_sync: Optional[str] = None):
        self.venv = venv
        self.device = venv.device
        self.num_envs, self.obs_dim = venv.num_envs, venv.obs_dim
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.obs_rms = RunningMeanStd((self.obs_dim,), self.device)
        self.ret_rms = RunningMeanStd((), self.device)
        self.returns = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
        self.obs_out = torch.zeros((self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device)
        self.tobs_out = torch.zeros_like(self.obs_out)
        # fused HIP passes (moments + merge + normalise) whenever the env lives on a GPU -- also in a sharded job, where
        # the kernels additionally accumulate the batch sums that sync_statistics() all-reduces once per rollout
        self.use_fused = (self.device.type == "cuda") if use_fused_kernel is None else use_fused_kernel
        self.stats_sync = ("rollout" if self.use_fused else "step") if stats_sync is None else stats_sync
        if self.stats_sync not in ("step", "rollout") or (self.use_fused and self.stats_sync == "step" and _dist() is not None):
            raise ValueError("stats_sync must be 'rollout' (fused kernels) or 'step' / 'rollout' (torch restatement)")
        self._ws = None
        self._obs_acc = self._ret_acc = None
        if self.use_fused:
            nbytes = int(_lib.lib().fw_normalize_obs_workspace_bytes(self.obs_dim))
            self._ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=self.device)       # caller-owned scratch of fw_normalize_obs
            nb2 = int(_lib.lib().fw_collect_stats_workspace_bytes(self.obs_dim))
            self._ws_stats = torch.zeros((nb2 + 7) // 8, dtype=torch.float64, device=self.device)   # ... and of fw_collect_stats (zeroed once)

        if _dist() is not None and self.stats_sync == "rollout":
            self._obs_acc = torch.zeros(2 * self.obs_dim + 1, dtype=torch.float64, device=self.device)
            self._ret_acc = torch.zeros(3, dtype=torch.float64, device=self.device)
            self._snap = self._snapshot()                   # the statistics every rank agreed on last (here: the initial ones)
        self.version = 0          # bumped whenever a scalar a captured rollout graph froze may have changed (load_state_dict, flag overrides)

    # -- observation ---------------------------------------------------------------------------
    def _norm_obs_torch(self, obs, out):
        z = (obs.to(torch.float64) - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + self.epsilon)
        out.copy_(z.to(torch.float32).clamp_(-self.clip_obs, self.clip_obs))
        return out

    def _process_obs(self, obs, update: bool):
        if not self.norm_obs:
            self.obs_out.copy_(obs.to(torch.float32))
            return self.obs_out
        if self.use_fused:
            rc = _lib.lib().fw_normalize_obs(_p(obs), int(obs.dtype == torch.float64), self.num_envs, self.obs_dim,
                                             _p(self.obs_rms.mean), _p(self.obs_rms.var), _p(self.obs_rms.count),
                                             int(update), float(self.clip_obs), float(self.epsilon), _p(self.obs_out),
                                             _p(self._ws), _p(self._obs_acc) if update else None, _stream(self.device))
            _lib.check(rc)
            return self.obs_out
        if update:
            self._update_rms(self.obs_rms, obs, self._obs_acc)
        return self._norm_obs_torch(obs, self.obs_out)

    # -- statistics of a sharded job ------------------------------------------------------------------
    def _update_rms(self, rms, x, acc):
        """Torch-path update: every vec-step over all ranks ("step"), or locally + accumulated for the per-rollout
        exchange ("rollout": same arithmetic as the fused kernels)."""
        if acc is None:
            rms.update(x)                                     # all-reduces the batch sums itself in a multi-process job
            return
        x = x.to(torch.float64).reshape(x.shape[0], -1)
        sx, sx2 = x.sum(0), (x * x).sum(0)
        n = torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)
        rms.update_from_sums(sx, sx2, n)
        acc.add_(torch.cat([sx, sx2, n]))

    def _snapshot(self):
        return [t.clone() for t in (self.obs_rms.mean, self.obs_rms.var, self.obs_rms.count,
                                    self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count)]

    # -- what a rollout moves, kept so that a void rollout (PPO.check_collect_status) can be taken back -------------------
    def _stat_tensors(self):
        live = [self.obs_rms.mean, self.obs_rms.var, self.obs_rms.count, self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count, self.returns]
        if self._obs_acc is not None:                        # sharded job: the batch sums since the last exchange belong to the state too
            live += [self._obs_acc, self._ret_acc]           # (not zero before the first rollout: the reset's observations are in them)
        return live

    def save_statistics(self) -> None:
        """In front of a rollout: one multi-tensor copy of the running statistics, the discounted-return accumulators and (sharded job)
        the batch sums not yet exchanged -- 2 D + 4 + N (+ 2 D + 4) doubles -- into persistent buffers."""
        live = self._stat_tensors()
        if getattr(self, "_stat_saved", None) is None:
            self._stat_saved = [torch.zeros_like(t) for t in live]
        torch._foreach_copy_(self._stat_saved, live)
        self._snap_saved = getattr(self, "_snap", None)      # (the agreed base of a sharded job: replaced at every exchange, never written in place)

    def restore_statistics(self) -> bool:
        """Back to what save_statistics() kept: the statistics (and, in a sharded job, the agreed base and the batch sums since) no
        longer contain anything of the rollout(s) in between.  False if nothing was saved."""
        if getattr(self, "_stat_saved", None) is None:
            return False
        torch._foreach_copy_(self._stat_tensors(), self._stat_saved)
        if self._obs_acc is not None:
            self._snap = self._snap_saved
        return True

    def sync_statistics(self) -> None:
        """Once per rollout in a sharded job (``stats_sync="rollout"``): all-reduce the accumulated batch sums (one small
        collective: 2 D + 4 doubles) and re-derive the running statistics as  statistics at the last sync  (+)  the batches
        of ALL ranks since -- afterwards every rank holds identical statistics, those of one VecNormalize over all envs."""
        if self._obs_acc is None:
            return
        d = self.obs_dim
        buf = all_reduce_sum_(torch.cat([self._obs_acc, self._ret_acc]))
        for rms, snap, (sx, sx2, n) in ((self.obs_rms, self._snap[0:3], (buf[:d], buf[d:2 * d], buf[2 * d:2 * d + 1])),
                                        (self.ret_rms, self._snap[3:6], (buf[2 * d + 1:2 * d + 2], buf[2 * d + 2:2 * d + 3], buf[2 * d + 3:]))):
            rms.mean.copy_(snap[0]); rms.var.copy_(snap[1]); rms.count.copy_(snap[2])
            if float(n.item()) > 0:
                rms.update_from_sums(sx, sx2, n)
        self._obs_acc.zero_(); self._ret_acc.zero_()
        self._snap = self._snapshot()

    def normalize_obs(self, obs, out=None):
        """Normalise with the current statistics without updating them."""
        out = torch.empty(obs.shape, dtype=torch.float32, device=obs.device) if out is None else out
        if not self.norm_obs:
            out.copy_(obs.to(torch.float32)); return out
        return self._norm_obs_torch(obs, out)

    # -- API -----------------------------------------------------------------------------------
    def reset(self):
        obs = self.venv.reset_tensor()
        self.returns.zero_()
        return self._process_obs(obs, update=self.training)

    def step(self, actions):
        obs, rew, term, trunc = self.venv.step_tensor(actions)
        dones = (term | trunc).bool()
        obs_n = self._process_obs(obs, update=self.training)
        rew64 = rew.to(torch.float64)
        if self.training and self.norm_reward:
            self.returns.mul_(self.gamma).add_(rew64)               # in place (graph-captured address)
            self._update_rms(self.ret_rms, self.returns, self._ret_acc)
        if self.norm_reward:
            rew_n = (rew64 / torch.sqrt(self.ret_rms.var + self.epsilon)).clamp(-self.clip_reward, self.clip_reward)
        else:
            rew_n = rew64
        tobs_n = self.normalize_obs(self.venv.terminal_obs, self.tobs_out)
        self.returns.masked_fill_(dones, 0.0)
        return obs_n, rew_n.to(torch.float32), dones, trunc.bool() & ~term.bool(), tobs_n

    def state_dict(self):
        return {"obs_rms": self.obs_rms.state_dict(), "ret_rms": self.ret_rms.state_dict(),
                "clip_obs": self.clip_obs, "clip_reward": self.clip_reward, "gamma": self.gamma,
                "epsilon": self.epsilon, "norm_obs": self.norm_obs, "norm_reward": self.norm_reward}

    def load_state_dict(self, sd):
        self.obs_rms.load_state_dict(sd["obs_rms"]); self.ret_rms.load_state_dict(sd["ret_rms"])
        if self._obs_acc is not None:                        # loaded statistics are the new synchronised base
            self._obs_acc.zero_(); self._ret_acc.zero_(); self._snap = self._snapshot()
        for k in ("clip_obs", "clip_reward", "gamma", "epsilon", "norm_obs", "norm_reward"):
            setattr(self, k, sd[k])
        self.version += 1


# ---------------------------------------------------------------------------------------------
# policy: SB3 "MlpPolicy" (ActorCriticPolicy defaults)
# ---------------------------------------------------------------------------------------------
class MlpPolicy(nn.Module):
    """Separate 64-64 tanh networks for pi and V, orthogonal init (gain sqrt(2) hidden, 0.01
    action head, 1 value head), state-independent log_std initialised to 0, diagonal Gaussian."""

    def __init__(self, obs_dim: int, act_dim: int = 4, hidden=(64, 64)):
        super().__init__()

        def mlp():
            layers, d = [], obs_dim
            for h in hidden:
                layers += [nn.Linear(d, h), nn.Tanh()]
                d = h
            return nn.Sequential(*layers), d

        self.pi_net, dpi = mlp()
        self.vf_net, dvf = mlp()
        self.action_net = nn.Linear(dpi, act_dim)
        self.value_net = nn.Linear(dvf, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))
        for net in (self.pi_net, self.vf_net):
            for m in net:
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2)); nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.action_net.weight, gain=0.01); nn.init.zeros_(self.action_net.bias)
        nn.init.orthogonal_(self.value_net.weight, gain=1.0); nn.init.zeros_(self.value_net.bias)

    def _dist_params(self, obs):
        return self.action_net(self.pi_net(obs)), self.log_std

    @staticmethod
    def _log_prob(actions, mean, log_std):
        var = torch.exp(2 * log_std)
        return (-((actions - mean) ** 2) / (2 * var) - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def forward(self, obs, deterministic: bool = False, generator=None):
        mean, log_std = self._dist_params(obs)
        if deterministic:
            actions = mean
        else:
            noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
            actions = mean + noise * torch.exp(log_std)
        values = self.value_net(self.vf_net(obs)).squeeze(-1)
        return actions, values, self._log_prob(actions, mean, log_std)

    def predict_values(self, obs):
        return self.value_net(self.vf_net(obs)).squeeze(-1)

    def evaluate_actions(self, obs, actions):
        mean, log_std = self._dist_params(obs)
        values = self.value_net(self.vf_net(obs)).squeeze(-1)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1).expand(obs.shape[0])
        return values, self._log_prob(actions, mean, log_std), entropy


class CnnDetectorPolicy(nn.Module):
    """MlpPolicy with a CNN "detector head" in front: a small conv net turns the FPV render of the env's analytic scene
    (``fw_render``: duck mask + depth buffer, float32 ``[N, 2, res, res]``) into ``cnn_features`` numbers that are
    concatenated to the flat observation; the 64-64 tanh policy / value networks, the Gaussian head and the initialisation
    are SB3's, as in :class:`MlpPolicy`; the extractor is shared by both networks like SB3's ``CnnPolicy`` shares its
    ``features_extractor``.  BASELINE.json configs[4] ("PPO with CNN detector head on PyTorch-ROCm"); the reference's own CNN
    path feeds camera images to a network (envs/fixedwing_envs/objlock_yolo_env.py:646-716) -- its trainers use ``MlpPolicy``
    (train/train_Fixedwing_Waypoints_ObjLock.py:349), so the architecture here is build-owned: conv 4x4 / 2 -> ReLU ->
    conv 3x3 / 2 -> ReLU -> conv 3x3 / 2 -> ReLU -> linear -> ReLU.  Plain torch-ROCm (MIOpen) -- no hand kernel."""

    uses_image = True

    def __init__(self, obs_dim: int, image_res: int = 32, act_dim: int = 4, hidden=(64, 64), cnn_features: int = 32, channels: int = 2):
        super().__init__()
        self.image_res, self.obs_dim = int(image_res), int(obs_dim)
        self.cnn = nn.Sequential(nn.Conv2d(channels, 16, 4, stride=2, padding=1), nn.ReLU(),
                                 nn.Conv2d(16, 32, 3, stride=2, padding=1), nn.ReLU(),
                                 nn.Conv2d(32, 32, 3, stride=2, padding=1), nn.ReLU(), nn.Flatten())
        with torch.no_grad():
            n_flat = self.cnn(torch.zeros(1, channels, self.image_res, self.image_res)).shape[1]
        self.cnn_head = nn.Sequential(nn.Linear(n_flat, cnn_features), nn.ReLU())
        self.mlp = MlpPolicy(obs_dim + cnn_features, act_dim, hidden)
        for m in list(self.cnn) + list(self.cnn_head):
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.orthogonal_(m.weight, gain=math.sqrt(2)); nn.init.zeros_(m.bias)

    @property
    def log_std(self):
        return self.mlp.log_std

    def image_features(self, img):
        """The extractor's output for a batch of images (callers that evaluate several heads on the same images compute it once)."""
        return self.cnn_head(self.cnn(img))

    def features(self, obs, img, feat=None):
        if feat is None:
            if img is None:
                raise ValueError("CnnDetectorPolicy needs the FPV image of the step (img=...)")
            feat = self.image_features(img)
        return torch.cat([obs, feat], dim=-1)

    def forward(self, obs, deterministic: bool = False, generator=None, img=None, feat=None):
        return self.mlp(self.features(obs, img, feat), deterministic=deterministic, generator=generator)

    def predict_values(self, obs, img=None, feat=None):
        return self.mlp.predict_values(self.features(obs, img, feat))

    def evaluate_actions(self, obs, actions, img=None, feat=None):
        return self.mlp.evaluate_actions(self.features(obs, img, feat), actions)


def policy_inputs(policy, env, out: Optional[torch.Tensor] = None) -> dict:
    """Extra inputs of a policy call besides the flat observation: ``{"img": FPV render of env's current state}`` for a policy
    with a CNN front end (``uses_image``), nothing otherwise."""
    if not getattr(policy, "uses_image", False):
        return {}
    return {"img": env.venv.render_tensor(policy.image_res, out=out)}


# ---------------------------------------------------------------------------------------------
# GAE
# ---------------------------------------------------------------------------------------------
def gae_reference(rewards, values, episode_starts, last_values, last_dones, gamma, lam):
    """Plain-torch restatement of SB3 RolloutBuffer.compute_returns_and_advantage (checker for fw_gae)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        if t == T - 1:
            nnt, nv = 1.0 - last_dones, last_values
        else:
            nnt, nv = 1.0 - episode_starts[t + 1], values[t + 1]
        delta = rewards[t] + gamma * nv * nnt - values[t]
        last = delta + gamma * lam * nnt * last
        adv[t] = last
    return adv, adv + values


def gae_device(rewards, values, episode_starts, last_values, last_dones, gamma, lam):
    """fw_gae: one lane per env, time loop in registers, [T, N] float32 device buffers."""
    if not rewards.is_cuda:
        raise RuntimeError("fw_gae needs device tensors (no CPU fallback); tests use gae_reference as the checker")
    T, N = rewards.shape
    adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
    rc = _lib.lib().fw_gae(_p(rewards), _p(values), _p(episode_starts), _p(last_values.contiguous()),
                           _p(last_dones.contiguous()), _p(adv), _p(ret), T, N, float(gamma), float(lam),
                           _stream(rewards.device))
    _lib.check(rc)
    return adv, ret


# ---------------------------------------------------------------------------------------------
# distributed helpers (update time only)
# ---------------------------------------------------------------------------------------------
def global_advantage_stats(adv: torch.Tensor):
    """Mean / std (unbiased, as torch.std) of the advantages of ALL ranks.

    north_star: an all-gather of the advantages over RCCL.  The payload is small
    (N_local x T x 4 B); the gathered tensor is also what a global minibatch sampler needs."""
    td = _dist()
    flat = adv.reshape(-1)
    if td is None:
        return flat.mean(), flat.std(), flat
    parts = [torch.empty_like(flat) for _ in range(td.get_world_size())]
    td.all_gather(parts, flat.contiguous())
    allv = torch.cat(parts)
    return allv.mean(), allv.std(), allv


def allreduce_grads_(params) -> None:
    """Average gradients over ranks with ONE flattened bucket (the MLP has ~1.2e4 parameters:
    latency-bound, so a single collective per minibatch)."""
    td = _dist()
    if td is None:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    td.all_reduce(flat)
    flat /= td.get_world_size()
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g)); o += n


# ---------------------------------------------------------------------------------------------
# PPO
# ---------------------------------------------------------------------------------------------
@dataclass
class PPOConfig:
    """TRAIN_CONFIG of train/train_Fixedwing_Waypoints_v3.py:27-55.  With thousands of envs
    ``n_steps`` is what shrinks: 4096 envs x 16 steps = the reference's 32 x 2048 = 65 536
    samples per update."""
    n_steps: int = 16
    batch_size: int = 128
    n_epochs: int = 20
    learning_rate: float = 3e-4
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    ent_coef: float = 0.001
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    normalize_advantage: bool = True
    adv_norm_scope: str = "minibatch"      # "minibatch" (SB3) | "global" (all-gathered statistics)
    seed: int = 42
    use_graphs: bool = True                # replay the rollout / minibatch update as hipGraphs (single-GPU, device envs)
    fused_update: bool = True              # run the whole minibatch sequence of train() in one HIP kernel (fw_ppo_update) when it applies
    fused_collect: bool = True             # policy forward / sampling / buffer writes and the reward path as fw_policy_act + fw_rollout_post
    one_launch_collect: bool = True        # the whole vec-step as ONE launch (fw_collect_step) where the env's lane mapping has it (8 lanes per env,
                                           # one wave per SIMD), fw_collect_act -> fw_step -> fw_collect_stats elsewhere or when False: 34.9 vs 47.9 us
                                           # per vec-step (waypoints, 4096 envs; profiles/r04_rollout_bench*.jsonl)
    collect_fallback: bool = True          # a fw_collect_step status error (a bounded in-grid wait ran out: the launch assumed a workgroup dispatch order
                                           # the platform does not promise) voids the rollout; True: take it back, re-arm on the three-launch
                                           # collector (fw_collect_act -> fw_step -> fw_collect_stats: no in-grid wait), warn once and keep
                                           # training; False: raise RuntimeError (SB3's "VecEnv.step returns or raises")
    detector: str = "none"                 # "cnn": CnnDetectorPolicy over the FPV render (fw_render) of a camera task -- torch path, gradient all-reduce
    image_res: int = 32                    #        side of the rendered image
    cnn_features: int = 32                 #        width of the extractor's output
    cnn_graphs: bool = True                #        replay the CNN rollout / minibatch step as hipGraphs (single process): update 2.4 -> 1.0 ms per 1024-sample minibatch
    dist_update: str = "replicated"        # multi-process job: "replicated" = all-gather the rollout shards, every rank runs the same
                                           # minibatch sequence (no per-minibatch collective); "allreduce" = local minibatches + gradient all-reduce


class _PpoHyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "clip_range", "ent_coef", "vf_coef", "max_grad_norm", "beta1", "beta2", "eps",
                                         "adv_mean", "adv_std")] + [("norm_adv", C.c_int32), ("step0", C.c_int32)]


class FusedPpoUpdate:
    """Host side of ``fw_ppo_update``: keeps the flat float32 images of the parameters and Adam moments the kernel
    works on (layout in include/fwsim.h) and moves them to / from the ``nn.Module`` and ``torch.optim.Adam`` state, so
    checkpoints, the torch path and the kernel stay interchangeable."""

    def __init__(self, policy: "MlpPolicy", optimizer: torch.optim.Adam, obs_dim: int):
        self.policy, self.opt, self.D = policy, optimizer, obs_dim
        self.Dp = (obs_dim + 1) & ~1
        dev = policy.log_std.device
        n = _lib.lib().fw_ppo_param_count(obs_dim)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)          # flat order (staging)
        ns = _lib.lib().fw_ppo_moment_count()
        smap = np.empty(ns, dtype=np.int32)
        _lib.check(_lib.lib().fw_ppo_moment_map(obs_dim, smap.ctypes.data_as(C.c_void_p)))
        owned = np.nonzero(smap >= 0)[0]
        self._slot = torch.as_tensor(owned, dtype=torch.long, device=dev)                  # owned slots ...
        self._flat_of_slot = torch.as_tensor(smap[owned], dtype=torch.long, device=dev)    # ... and their flat indices
        self.mom_m = torch.zeros(ns, dtype=torch.float32, device=dev)                      # slot order (what the kernel sees)
        self.mom_v = torch.zeros_like(self.mom_m)
        self.loss = torch.zeros(16, dtype=torch.float32, device=dev)
        self._ws = None                    # caller-owned scratch of fw_ppo_update (grown here, never inside the call)
        self.synced = False                # flat / mom_m / mom_v are what the module and the optimiser hold (set by commit(); whoever changes
        self._step = 0                     # either behind this object's back clears it: PPO does wherever it clears _flat_current)
        self.last_paths = 0                # fw_ppo_update_status: which exchanges of the last call went through a shared L2
        self._sig = None                   # state_signature() of module + optimiser when the images were last known equal to them (commit())
        self._param_sig = None             # param_signature() when `flat` was last loaded from / stored to the module

    def _workspace(self, n_mb: int, batch_size: int) -> torch.Tensor:
        # exchange words + gradient hand-off buffer + the packed rows of every minibatch (a parallel pre-pass of the call writes them)
        need = int(_lib.lib().fw_ppo_update_workspace_bytes(n_mb, batch_size, self.D))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.flat.device)
        return self._ws

    @staticmethod
    def fits(policy, obs_dim: int, device) -> bool:
        """The kernels are written for the reference's MlpPolicy: two 64-64 tanh nets, 4 actions, obs_dim <= 64 (any number
        of ranks: a sharded job runs them unchanged on every GPU)."""
        if getattr(policy, "uses_image", False) or not hasattr(policy, "pi_net"):
            return False                   # CNN front end: torch path (the fused kernels are the MlpPolicy's)
        lin = [m for m in list(policy.pi_net) + list(policy.vf_net) if isinstance(m, nn.Linear)]
        return (device.type == "cuda" and obs_dim <= 64 and len(lin) == 4
                and all(m.out_features == 64 for m in lin) and policy.action_net.out_features == 4)

    @staticmethod
    def applies(policy, cfg, obs_dim: int, batch_size: int, device) -> bool:
        return cfg.fused_update and batch_size % 16 == 0 and FusedPpoUpdate.fits(policy, obs_dim, device)

    def _slots(self):
        """(tensor, flat offset, view shape in the flat image, needs transpose) per parameter, in layout order."""
        p, out, off = self.policy, [], 0
        for net, head in ((p.pi_net, p.action_net), (p.vf_net, p.value_net)):
            l1, l2 = net[0], net[2]
            ko = head.out_features
            out.append((l1.weight, off, (self.Dp, 64), True)); off += self.Dp * 64
            out.append((l1.bias, off, (64,), False)); off += 64
            out.append((l2.weight, off, (64, 64), True)); off += 64 * 64
            out.append((l2.bias, off, (64,), False)); off += 64
            out.append((head.weight, off, (64, ko), True)); off += 64 * ko
            out.append((head.bias, off, (ko,), False)); off += ko
        out.append((p.log_std, off, (4,), False)); off += 4
        assert off == self.flat.numel()
        return out

    def _put(self, flat, src, off, shape, tr):
        view = flat[off:off + int(np.prod(shape))].view(shape)
        if tr:
            view[:src.shape[1], :].copy_(src.t())         # W[in][out] = weight^T (rows past obs_dim stay zero)
        else:
            view.copy_(src)

    def _get(self, flat, dst, off, shape, tr):
        view = flat[off:off + int(np.prod(shape))].view(shape)
        dst.copy_(view[:dst.shape[1], :].t() if tr else view)

    # The images are caches of the module / optimiser.  Whoever writes those tensors in place (``p.add_()``,
    # ``policy.load_state_dict``, an edit of ``exp_avg``, ``optimizer.step()``) bumps their ``_version``; whoever replaces them
    # (``optimizer.load_state_dict``) changes their address: the signatures below see both, and a mismatch reloads the image
    # instead of trusting a flag.  (Writes through ``tensor.data`` bypass the version counter by design of torch: callers that
    # do that say so with ``PPO.touch()``.)
    def param_signature(self):
        return tuple((t.data_ptr(), t._version) for t, _, _, _ in self._slots())

    def state_signature(self):
        sig = list(self.param_signature())
        if self.opt is not None:
            for t, _, _, _ in self._slots():
                st = self.opt.state.get(t)
                if not st:
                    sig.append(None); continue
                for k in ("exp_avg", "exp_avg_sq", "step"):
                    v = st.get(k)
                    sig.append((v.data_ptr(), v._version) if torch.is_tensor(v) else (k, v))
        return tuple(sig)                  # (the hyper-parameters of the optimiser are read at every call: nothing of them is cached)

    def params_current(self) -> bool:
        return self._param_sig is not None and self._param_sig == self.param_signature()

    @torch.no_grad()
    def load_params_from_torch(self) -> None:
        for t, off, shape, tr in self._slots():
            self._put(self.flat, t.data, off, shape, tr)
        self._param_sig = self.param_signature()

    @torch.no_grad()
    def load_from_torch(self) -> int:
        self.flat.zero_(); self.m.zero_(); self.v.zero_()
        step = 0
        for t, off, shape, tr in self._slots():
            self._put(self.flat, t.data, off, shape, tr)
            st = self.opt.state.get(t)
            if st:
                self._put(self.m, st["exp_avg"], off, shape, tr); self._put(self.v, st["exp_avg_sq"], off, shape, tr)
                step = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
        self.mom_m.zero_(); self.mom_v.zero_()
        self.mom_m[self._slot] = self.m[self._flat_of_slot]; self.mom_v[self._slot] = self.v[self._flat_of_slot]
        return step

    @torch.no_grad()
    def store_to_torch(self, step: int) -> None:
        capturable = bool(self.opt.param_groups[0].get("capturable", False))
        self.m[self._flat_of_slot] = self.mom_m[self._slot]; self.v[self._flat_of_slot] = self.mom_v[self._slot]
        for t, off, shape, tr in self._slots():
            self._get(self.flat, t.data, off, shape, tr)
            st = self.opt.state[t]
            if not st:
                st["step"] = torch.zeros((), dtype=torch.float32, device=t.device if capturable else "cpu")
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(t), torch.zeros_like(t)
            self._get(self.m, st["exp_avg"], off, shape, tr); self._get(self.v, st["exp_avg_sq"], off, shape, tr)
            if torch.is_tensor(st["step"]):
                st["step"].fill_(float(step))
            else:
                st["step"] = step

    def run(self, cfg, obs, act, old_logp, adv, ret, perm_i32, n_mb: int, g_mean: float, g_std: float):
        # (the images of the previous call are still current when nothing else touched the module / optimiser: ~70 small copies and a
        # host sync less per update)
        step0 = self._step if (self.synced and self._sig == self.state_signature()) else self.load_from_torch()
        self.synced = False                # until commit(): the kernel is about to move the images
        self._param_sig = None
        pg = self.opt.param_groups[0]
        H = _PpoHyper(lr=pg["lr"], clip_range=cfg.clip_range, ent_coef=cfg.ent_coef, vf_coef=cfg.vf_coef,
                      max_grad_norm=cfg.max_grad_norm, beta1=pg["betas"][0], beta2=pg["betas"][1], eps=pg["eps"],
                      adv_mean=g_mean, adv_std=g_std,
                      norm_adv=(0 if not cfg.normalize_advantage else (2 if cfg.adv_norm_scope == "global" else 1)), step0=step0)
        self.loss.zero_()
        for x in (obs, act, old_logp, adv, ret):
            assert x.dtype == torch.float32 and x.is_contiguous()
        assert perm_i32.dtype == torch.int32 and perm_i32.numel() == n_mb * cfg.batch_size
        ws = self._workspace(n_mb, cfg.batch_size)
        rc = _lib.lib().fw_ppo_update(_p(self.flat), _p(self.mom_m), _p(self.mom_v), _p(obs), _p(act), _p(old_logp), _p(adv), _p(ret),
                                      _p(perm_i32), n_mb, cfg.batch_size, self.D, C.byref(H), _p(self.loss), _p(ws), ws.numel(),
                                      _stream(obs.device))
        _lib.check(rc)
        # the workgroups of the launch wait for each other, every wait bounded: a wait that ran out leaves a status word behind and
        # (unless the closing verdict itself was lost) untouched images -- surface it BEFORE anything is written back to the module /
        # optimiser (they still hold the state of before the call)
        st, paths = C.c_uint32(0), C.c_uint32(0)
        _lib.check(_lib.lib().fw_ppo_update_status(_p(ws), ws.numel(), C.byref(st), C.byref(paths), _stream(obs.device)))
        self.last_paths = int(paths.value)
        if st.value:
            names = [n for b, n in ((1, "block ids"), (2, "gradient swap"), (4, "norm exchange"), (8, "closing verdict")) if st.value & b]
            # (the flat images are undefined after a non-zero status -- include/fwsim.h -- and are reloaded at the next call; the
            # module and the optimiser, which only commit() writes, still hold the state of before the call)
            raise RuntimeError(f"fw_ppo_update gave up inside the launch (status {st.value}: {', '.join(names)} wait ran out); "
                               "the policy and optimiser were left as they were before the call")
        self._pending_step = step0 + n_mb      # commit() moves the result into the module / optimiser
        return (self.loss[:3] / n_mb).tolist()

    def commit(self) -> None:
        """Second half of an update: the new parameters and moments go from the flat images to the module / optimiser.  Kept apart
        from run() so that a sharded job can first agree that the launch ran to its end on EVERY rank."""
        self.store_to_torch(self._pending_step)
        self._step, self.synced = self._pending_step, True
        self._sig = self.state_signature()          # (taken AFTER the store: its copies moved the counters)
        self._param_sig = self.param_signature()


class PPO:
    def __init__(self, env: VecNormalizeDevice, cfg: PPOConfig = PPOConfig(), policy: Optional[MlpPolicy] = None,
                 gae_fn=gae_device):
        self.env, self.cfg, self.device = env, cfg, env.device
        self._gae = gae_fn
        torch.manual_seed(cfg.seed)
        if cfg.detector not in ("none", "cnn"):
            raise ValueError("detector must be 'none' or 'cnn'")
        if policy is None and cfg.detector == "cnn":
            policy = CnnDetectorPolicy(env.obs_dim, image_res=cfg.image_res, cnn_features=cfg.cnn_features)
        self.policy = (policy or MlpPolicy(env.obs_dim)).to(self.device)
        self._img = bool(getattr(self.policy, "uses_image", False))
        if self._img and not hasattr(env.venv, "render_tensor"):
            raise ValueError("a policy with a CNN front end needs an env with render_tensor() (a camera task on the device)")
        td = _dist()
        if td is not None:                 # identical initial weights on every rank
            for p in self.policy.parameters():
                td.broadcast(p.data, src=0)
        if cfg.dist_update not in ("replicated", "allreduce"):
            raise ValueError("dist_update must be 'replicated' or 'allreduce'")
        # CNN front end: data-parallel minibatches + ONE flattened gradient all-reduce each (the images of a rollout are ~30x
        # the flat observations: gathering every rank's rollout on every rank is the wrong trade there)
        self._replicated = td is not None and cfg.dist_update == "replicated" and not self._img
        self._fused_collect_ok = (bool(cfg.fused_collect) and FusedPpoUpdate.fits(self.policy, env.obs_dim, self.device)
                                  and getattr(env, "use_fused", False) and env.norm_obs and hasattr(env.venv, "step_tensor")
                                  and hasattr(env.venv, "terminal_obs") and hasattr(env.venv, "torch_dtype"))
        # hipGraph replay needs a collective-free body: always on one GPU; in a sharded job when the collector is fused
        # (its statistics are exchanged BETWEEN rollouts) and the update is replicated (no gradient all-reduce)
        # (CNN front end: its convolutions are captured like everything else when PPOConfig.cnn_graphs is on -- MIOpen picks its
        # algorithm in the eager warm-up rollout / minibatches that precede every capture)
        self._graphs = (bool(cfg.use_graphs) and self.device.type == "cuda" and (not self._img or bool(cfg.cnn_graphs))
                        and (td is None or (self._fused_collect_ok and self._replicated)))
        self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=cfg.learning_rate, eps=1e-5,
                                          capturable=self._graphs)
        self._g_rollout = self._g_update = None
        self._g_rollout_key = None
        self._fused = None
        self._flat_current = False         # does self._fused.flat hold the current policy parameters?
        self._collect_fused = self._fused_collect_ok
        if self._collect_fused:
            self._fused = FusedPpoUpdate(self.policy, self.optimizer, env.obs_dim)
            self._rng = torch.tensor([cfg.seed * 7919 + 17, 0], dtype=torch.int64, device=self.device)      # seed, draw counter
            # (NaN = "not there yet": fw_collect_step's step waves see their actions replace it and put it back)
            self._act_env = torch.full((env.num_envs, 4), float("nan"), dtype=env.venv.torch_dtype, device=self.device)
            self._tval = torch.zeros(env.num_envs, dtype=torch.float32, device=self.device)
        # one launch per vec-step (fw_collect_step) where the handle's lane mapping has it: 8 lanes per env (either build)
        # (fw_collect_step's act waves take observations of up to 62 features -- wider ones go through fw_collect_act, which takes 64)
        self._one_launch = (self._collect_fused and bool(cfg.one_launch_collect) and hasattr(env.venv, "_h") and env.obs_dim <= 62
                            and getattr(env.venv, "lanes_per_env", 0) == 8 and float(env.gamma) == float(cfg.gamma))
        self._void_recoverable = False
        self.collect_fallbacks = 0         # how many times a void rollout moved this object to the three-launch collector (0 or 1)
        self._void_steps = 0               # timesteps the last rollout added to num_timesteps (taken back if it turns out void)
        self._ws_collect = None
        self._status_host = None           # CS_STATUS of the workspace, copied to pinned memory as the last command of every rollout
        self._status_pending = False
        # ... and then the rollout ends in one launch too (fw_collect_close), unless the caller brought its own GAE
        self._close_gae = bool(self._one_launch and gae_fn is gae_device)
        self._adv_buf = self._ret_buf = None
        self._trace = None                 # optional int64 [grid, 8] device tensor: per-workgroup wall-clock stamps of the last fw_collect_step
        self._warm_rollouts = 0
        self._gathered = None
        self.allgather_ms = self.allgather_bytes = 0.0
        self._loss_acc = torch.zeros(3, device=self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(cfg.seed + (td.get_rank() if td is not None else 0))
        # minibatch permutations: the SAME stream on every rank, so that a replicated update walks identical minibatches
        self.perm_gen = self.gen
        if self._replicated:
            self.perm_gen = torch.Generator(device=self.device)
            self.perm_gen.manual_seed(cfg.seed * 1_000_003 + 12345)
        T, N, D = cfg.n_steps, env.num_envs, env.obs_dim
        f32 = dict(dtype=torch.float32, device=self.device)
        self.buf_obs = torch.zeros((T, N, D), **f32)
        self.buf_act = torch.zeros((T, N, 4), **f32)
        self.buf_rew = torch.zeros((T, N), **f32)
        self.buf_start = torch.zeros((T, N), **f32)
        self.buf_val = torch.zeros((T, N), **f32)
        self.buf_logp = torch.zeros((T, N), **f32)
        if self._close_gae:
            self._adv_buf, self._ret_buf = torch.zeros((T, N), **f32), torch.zeros((T, N), **f32)
        self.buf_img = self.last_img = None
        if self._img:
            r = int(self.policy.image_res)
            self.buf_img = torch.zeros((T, N, 2, r, r), **f32)
            self.last_img = torch.zeros((N, 2, r, r), **f32)
        self.last_obs = None
        self.last_starts = torch.ones(N, **f32)
        self.num_timesteps = 0
        self.logs: Dict[str, float] = {}

    # ---- SB3 OnPolicyAlgorithm.collect_rollouts ---------------------------------------------
    def _rollout_body(self):
        cfg, env = self.cfg, self.env
        act_dtype = env.venv.torch_dtype
        kw = {}
        for t in range(cfg.n_steps):
            if self._img:
                # the FPV image of the state the policy acts on (fw_render of the env's current pose, refreshed below after the step)
                # (the conv extractor runs ONCE per step: the policy forward and the bootstrap value below see the same image)
                kw = {"feat": self.policy.image_features(self.last_img)}
                self.buf_img[t].copy_(self.last_img)
            actions, values, logp = self.policy(self.last_obs, generator=self.gen, **kw)
            clipped = actions.clamp(-1.0, 1.0).to(act_dtype)
            obs_n, rew_n, dones, timeouts, tobs_n = env.step(clipped)
            # bootstrap truncated episodes with V(terminal_observation).  (CNN front end: the env has already auto-reset, so the
            # terminal pose can no longer be rendered; the image of the step before stands in -- one agent step stale, and only
            # for the rare episodes that end on the time limit.)
            tv = self.policy.predict_values(tobs_n, **kw)
            rew_n = rew_n + cfg.gamma * tv * timeouts.to(torch.float32)
            self.buf_obs[t].copy_(self.last_obs); self.buf_act[t].copy_(actions); self.buf_rew[t].copy_(rew_n)
            self.buf_start[t].copy_(self.last_starts); self.buf_val[t].copy_(values); self.buf_logp[t].copy_(logp)
            self.last_obs.copy_(obs_n)
            self.last_starts.copy_(dones.to(torch.float32))
            if self._img:
                env.venv.render_tensor(self.policy.image_res, out=self.last_img)
        if self._img:
            kw = {"img": self.last_img}
        self.last_values.copy_(self.policy.predict_values(self.last_obs, **kw))

    def _act(self, obs, nets, t=None, value_out=None):
        L, env = _lib.lib(), self.env
        bo = _p(self.buf_obs[t]) if t is not None else None
        ba = _p(self.buf_act[t]) if t is not None else None
        bl = _p(self.buf_logp[t]) if t is not None else None
        val = value_out if value_out is not None else (self.buf_val[t] if t is not None else None)
        _lib.check(L.fw_policy_act(_p(self._fused.flat), _p(obs), env.num_envs, env.obs_dim, nets, 0, _p(self._rng),
                                   int(getattr(env.venv, "global_env_offset", 0)), bo, ba, _p(self._act_env),
                                   int(self._act_env.dtype == torch.float64), bl, _p(val), _stream(self.device)))

    def _rollout_body_fused(self):
        """collect_rollouts as THREE launches per vec-step:  fw_collect_act -> fw_step -> fw_collect_stats.
        fw_collect_act reads the env's raw observation buffer, normalises it on load with the current statistics (the policy
        sees what VecNormalize.step_wait would have returned), samples, writes the rollout-buffer rows of step t and the env's
        action -- and its value block first finalises step t-1 for its rows (normalised + bootstrapped reward, episode starts),
        while the env's output buffers still hold that step.  fw_collect_stats folds the step's observations and rewards into
        both normalisers.  Same data flow as _rollout_body; SB3 OnPolicyAlgorithm.collect_rollouts + VecNormalize semantics."""
        cfg, env, L = self.cfg, self.env, _lib.lib()
        venv, T, N, D = env.venv, cfg.n_steps, env.num_envs, env.obs_dim
        st = _stream(self.device)
        f64 = int(venv.obs.dtype == torch.float64)
        self.buf_start[0].copy_(self.last_starts)
        track = int(env.training and env.norm_reward)

        def act(t, nets, value_out, prev_t):
            """prev_t: index of the step to finalise (its rewards go to buf_rew[prev_t], the starts of the NEXT step to
            buf_start[prev_t + 1] / last_starts), or None."""
            bo = _p(self.buf_obs[t]) if t is not None else None
            ba = _p(self.buf_act[t]) if t is not None else None
            bl = _p(self.buf_logp[t]) if t is not None else None
            if prev_t is None:
                prev = (None, None, None, None, None, 0, 0.0, 0.0, 0.0, None, None)
            else:
                nxt = self.buf_start[prev_t + 1] if prev_t + 1 < T else self.last_starts
                prev = (_p(venv.rewards), _p(venv.terminated), _p(venv.truncated), _p(venv.terminal_obs), _p(env.ret_rms.var),
                        int(env.norm_reward), float(env.clip_reward), float(env.epsilon), float(cfg.gamma), _p(self.buf_rew[prev_t]), _p(nxt))
            _lib.check(L.fw_collect_act(_p(self._fused.flat), _p(venv.obs), f64, N, D, _p(env.obs_rms.mean), _p(env.obs_rms.var),
                                        float(env.clip_obs), float(env.epsilon), nets, 0, _p(self._rng),
                                        int(getattr(venv, "global_env_offset", 0)), bo, ba, _p(self._act_env),
                                        int(self._act_env.dtype == torch.float64), bl, _p(value_out), *prev, st))

        if self._one_launch:
            # ONE launch per vec-step (fw_collect_step): act waves, the env's step waves and the statistics fold share a grid
            from . import config as K
            if self._ws_collect is None:
                nb = int(L.fw_collect_step_workspace_bytes(venv._h))
                self._ws_collect = torch.empty((nb + 7) // 8, dtype=torch.float64, device=self.device)      # caller-owned, initialised once
                _lib.check(L.fw_collect_workspace_init(venv._h, self._ws_collect.data_ptr(), self._ws_collect.numel() * 8, st), venv._h)
            upd_obs = int(env.training and env.norm_obs)
            for t in range(T):
                a = K.FwCollectArgs()
                a.params = self._fused.flat.data_ptr()
                a.obs_mean, a.obs_var, a.obs_count = env.obs_rms.mean.data_ptr(), env.obs_rms.var.data_ptr(), env.obs_rms.count.data_ptr()
                a.returns = env.returns.data_ptr()
                a.ret_mean, a.ret_var, a.ret_count = env.ret_rms.mean.data_ptr(), env.ret_rms.var.data_ptr(), env.ret_rms.count.data_ptr()
                a.obs_acc = env._obs_acc.data_ptr() if (upd_obs and env._obs_acc is not None) else None
                a.ret_acc = env._ret_acc.data_ptr() if (track and env._ret_acc is not None) else None
                a.rng = self._rng.data_ptr()
                a.obs_copy, a.act_raw, a.logp, a.value = (self.buf_obs[t].data_ptr(), self.buf_act[t].data_ptr(), self.buf_logp[t].data_ptr(),
                                                          self.buf_val[t].data_ptr())
                a.act_env = self._act_env.data_ptr()
                if t > 0:                                        # the value waves finalise step t - 1 while its outputs are still in the env's buffers
                    a.rew_out, a.start_out = self.buf_rew[t - 1].data_ptr(), self.buf_start[t].data_ptr()
                a.obs, a.reward = venv.obs.data_ptr(), venv.rewards.data_ptr()
                a.terminated, a.truncated = venv.terminated.data_ptr(), venv.truncated.data_ptr()
                a.terminal_obs, a.info_i32 = venv.terminal_obs.data_ptr(), venv.info.data_ptr()
                a.workspace, a.workspace_bytes = self._ws_collect.data_ptr(), self._ws_collect.numel() * 8
                a.trace = self._trace.data_ptr() if self._trace is not None else None      # (tools/trace_collect.py)
                a.gamma = float(cfg.gamma)
                a.clip_obs, a.eps_obs, a.clip_reward, a.eps_reward = float(env.clip_obs), float(env.epsilon), float(env.clip_reward), float(env.epsilon)
                a.update_obs, a.update_ret, a.norm_reward, a.deterministic = upd_obs, track, int(env.norm_reward), 0
                _lib.check(L.fw_collect_step(venv._h, C.byref(a), st), venv._h)
            if self._close_gae:
                # ... and the end of the rollout in one more: the last step's statistics, V(last observation), the finalisation of
                # step T - 1, the normalised last observation and the GAE scan (fw_collect_close)
                a.obs_copy, a.value = self.last_obs.data_ptr(), self.last_values.data_ptr()
                a.act_raw = a.logp = a.act_env = None
                a.rew_out, a.start_out = self.buf_rew[T - 1].data_ptr(), self.last_starts.data_ptr()
                c = K.FwCollectCloseArgs()
                c.rewards, c.values, c.episode_starts = self.buf_rew.data_ptr(), self.buf_val.data_ptr(), self.buf_start.data_ptr()
                c.adv, c.ret, c.T = self._adv_buf.data_ptr(), self._ret_buf.data_ptr(), T
                c.gae_gamma, c.gae_lambda = float(cfg.gamma), float(cfg.gae_lambda)
                _lib.check(L.fw_collect_close(venv._h, C.byref(a), C.byref(c), st), venv._h)
                self._queue_collect_status()
                return
            _lib.check(L.fw_collect_finish(venv._h, C.byref(a), st), venv._h)      # the last step's statistics (each step's are merged by the next launch)
            self._queue_collect_status()
        for t in (range(T) if not self._one_launch else ()):
            act(t, 3, self.buf_val[t], t - 1 if t > 0 else None)
            venv.step_tensor(self._act_env)
            _lib.check(L.fw_collect_stats(_p(venv.obs), f64, N, D, _p(env.obs_rms.mean), _p(env.obs_rms.var), _p(env.obs_rms.count),
                                          int(env.training and env.norm_obs), _p(venv.rewards), int(venv.rewards.dtype == torch.float64),
                                          _p(venv.terminated), _p(venv.truncated), _p(env.returns), _p(env.ret_rms.mean),
                                          _p(env.ret_rms.var), _p(env.ret_rms.count), track, float(env.gamma), _p(self._rng),
                                          _p(env._ws_stats), _p(env._obs_acc) if env.training and env.norm_obs else None,
                                          _p(env._ret_acc) if track else None, st))
        # V(last observation) for GAE + the finalisation of step T-1 (nothing is sampled; the normalised last observation lands
        # in last_obs for callers that look at it)
        _lib.check(L.fw_collect_act(_p(self._fused.flat), _p(venv.obs), f64, N, D, _p(env.obs_rms.mean), _p(env.obs_rms.var),
                                    float(env.clip_obs), float(env.epsilon), 2, 0, _p(self._rng), int(getattr(venv, "global_env_offset", 0)),
                                    None, None, None, 0, None, _p(self.last_values),
                                    _p(venv.rewards), _p(venv.terminated), _p(venv.truncated), _p(venv.terminal_obs), _p(env.ret_rms.var),
                                    int(env.norm_reward), float(env.clip_reward), float(env.epsilon), float(cfg.gamma),
                                    _p(self.buf_rew[T - 1]), _p(self.last_starts), st))
        _lib.check(L.fw_normalize_obs(_p(venv.obs), f64, N, D, _p(env.obs_rms.mean), _p(env.obs_rms.var), _p(env.obs_rms.count), 0,
                                      float(env.clip_obs), float(env.epsilon), _p(self.last_obs), None, None, st))

    # ---- fw_collect_step's status word: "step returns or raises" ------------------------------
    _COLLECT_STATUS_BITS = ((1, "a step wave gave up waiting for its actions"), (2, "a fold wave summed without every partial sum"),
                            (4, "the merge wave committed before every act wave had read the old statistics"),
                            (8, "the workspace was never initialised"), (16, "the merge wave never saw the launch index"),
                            (32, "the policy produced a NaN action (diverged weights or statistics)"))

    _COLLECT_STATUS_WAITS = 1 | 2 | 4 | 16

    def _queue_collect_status(self) -> None:
        """Last command of a rollout through fw_collect_step (inside the captured graph when there is one): the workspace's status
        word goes to pinned host memory -- 4 bytes behind the closing launch, nothing between two replays (an event + side-stream
        copy per rollout cost 19 us of gaps per 16-step rollout).  train() / state_dict() / the next collect_rollouts() look at it."""
        if not self._one_launch or self._ws_collect is None:
            return
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_host.copy_(self._ws_collect.view(torch.int32)[-16 + 3:-16 + 4], non_blocking=True)

    def _collect_status_word(self, wait: bool, collective: bool) -> Tuple[int, int]:
        """(this rank's word, union over the ranks' words) of the rollout(s) since the last look; (0, 0) when there is nothing to
        look at (yet).  Consumes the pending flag when it looks."""
        td = _dist() if (collective and self._one_launch) else None      # (the same configuration on every rank: all or none take this branch)
        if td is None and (not self._status_pending or self._status_host is None):
            return 0, 0
        st = 0
        if self._status_pending and self._status_host is not None:
            stream = torch.cuda.current_stream(self.device)
            if wait:
                stream.synchronize()
            elif not stream.query():
                return 0, 0
            self._status_pending = False
            st = int(self._status_host.item())
        st_all = st
        if td is not None:
            bits = torch.tensor([float(bool(st & b)) for b, _ in self._COLLECT_STATUS_BITS], dtype=torch.float64, device=self.device)
            counts = all_reduce_sum_(bits).tolist()
            st_all = sum(b for (b, _), c in zip(self._COLLECT_STATUS_BITS, counts) if c > 0)
        return st, st_all

    def _take_back_void_rollout(self, st: int, st_all: int) -> str:
        """A launch of the last rollout went on without something it waited for: nothing collected since may be used.  Leaves a
        usable object behind -- fresh workspace and action words, no captured graph over the old ones, the normalisers' statistics
        and return accumulators as they were when the rollout began, no advantages for train() to pick up, the timestep counter
        without the void steps -- and, unless PPOConfig.collect_fallback is off, re-arms on the three-launch collector, which has no
        in-grid wait to run out.  Returns the message for the caller to raise or to log."""
        why = "; ".join(t for b, t in self._COLLECT_STATUS_BITS if st_all & b)
        where = "" if st == st_all else f" (another rank of the job reported status word {st_all & ~st}; this rank's own word: {st})"
        self._ws_collect = None
        self._g_rollout = None
        self._warm_rollouts = 0
        self._act_env.fill_(float("nan"))
        restored = self.env.restore_statistics() if hasattr(self.env, "restore_statistics") else False
        self.adv = self.ret = None
        self.num_timesteps -= self._void_steps
        self._void_steps = 0
        msg = (f"fw_collect_step: status word {st_all}{where} ({why}); the rollout is void -- its buffers are dropped"
               + (", the normaliser statistics are those of before it" if restored else ""))
        # bits 1 / 2 / 4 / 16 are waits of one wave for another that ran out -- what the other collector cannot have; a workspace that
        # was never initialised (8) or a policy that emits NaN (32) would fail there just the same: those always raise
        self._void_recoverable = bool(self.cfg.collect_fallback) and not (st_all & ~self._COLLECT_STATUS_WAITS)
        if self._void_recoverable:
            self._one_launch = self._close_gae = False
            self.collect_fallbacks += 1
            msg += "; re-armed on the three-launch collector (fw_collect_act -> fw_step -> fw_collect_stats)"
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return msg

    def check_collect_status(self, wait: bool = True, collective: bool = False) -> None:
        """Raise RuntimeError if a wait inside a fw_collect_step launch of the last rollout(s) ran out (the launch then went on
        with zero actions / partial statistics: everything collected since is void).  SB3's contract for ``VecEnv.step`` is
        "returns or raises"; a pipe to a dead SubprocVecEnv worker raises there
        (train/train_Fixedwing_Waypoints_v3.py:251, :332-337).  ``wait=False`` only looks if the stream has drained (no
        synchronisation).  ``collective=True`` (a point every rank of a sharded job reaches: train()): the ranks agree on the union
        of their words, so that all of them raise together instead of one leaving the others in the next collective -- which is
        why, in a sharded job, ONLY the collective look consumes the word: a rank-local look (collect_rollouts, state_dict,
        checkpoint.snapshot -- rank 0 alone, typically) that raised would strand the other ranks.
        Either way the object is usable afterwards (_take_back_void_rollout), on the three-launch collector unless
        ``PPOConfig.collect_fallback`` is off; learn() / train() do not raise at all then but collect the rollout again."""
        if not collective and self._one_launch and _dist() is not None:
            return
        st, st_all = self._collect_status_word(wait, collective)
        if st_all:
            raise RuntimeError(self._take_back_void_rollout(st, st_all))

    def touch(self) -> None:
        """Tell the object that module parameters or optimiser state were changed in a way torch's version counters do not see
        (writes through ``tensor.data``): collector and update reload their images.  In-place writes (``p.add_()``,
        ``load_state_dict``, edits of ``exp_avg``) and replaced tensors are noticed without this."""
        self._flat_current = False

    @torch.no_grad()
    def collect_rollouts(self):
        cfg, env = self.cfg, self.env
        if not (self._one_launch and _dist() is not None):      # (sharded: train()'s collective look only, see check_collect_status)
            st, st_all = self._collect_status_word(wait=False, collective=False)      # the previous rollout's word, if its copy has arrived
            if st_all:
                msg = self._take_back_void_rollout(st, st_all)
                if not self._void_recoverable:
                    raise RuntimeError(msg)
        if self.last_obs is None:
            self.last_obs = env.reset().clone()
            if self._img:
                env.venv.render_tensor(self.policy.image_res, out=self.last_img)
            self.last_starts.fill_(1.0)
            self.last_values = torch.zeros(env.num_envs, dtype=torch.float32, device=self.device)
        if self._one_launch and not self._status_pending:
            # every earlier rollout has been looked at: this is what a void rollout is taken back to (one multi-tensor copy in front of
            # the rollout; rollouts queued behind an unread word keep the older base, so that ALL of them can be taken back)
            if hasattr(env, "save_statistics"):
                env.save_statistics()
            self._void_steps = 0
        body = self._rollout_body
        if self._collect_fused:
            if not self._flat_current or not self._fused.params_current():      # (flag: PPO's own moves; signature: everybody else's)
                self._fused.load_params_from_torch(); self._flat_current = True
            body = self._rollout_body_fused
        key = (getattr(env, "version", 0), bool(env.training), bool(env.norm_reward), bool(env.norm_obs))
        if self._g_rollout is not None and key != self._g_rollout_key:
            self._g_rollout = None             # the graph froze clip / gamma / training flags as kernel arguments: re-capture
        self._g_rollout_key = key
        if self._graphs and self._warm_rollouts >= 1:
            # the n_steps x (policy forward, fw_step, fw_normalize_obs, buffer writes) chain is one
            # hipGraph: ~40 tiny launches per vec-step are otherwise host-bound (0.75 ms vs 27 us of physics).
            # The first rollout runs eagerly (it doubles as the allocator warm-up); capture happens on the second.
            if self._g_rollout is None:
                torch.cuda.synchronize()
                self._g_rollout = torch.cuda.CUDAGraph()
                if hasattr(self._g_rollout, "register_generator_state"):
                    self._g_rollout.register_generator_state(self.gen)
                with torch.cuda.graph(self._g_rollout):
                    body()
            self._g_rollout.replay()
        else:
            body()
        self._warm_rollouts += 1
        self._status_pending = self._one_launch
        if hasattr(env, "sync_statistics"):
            env.sync_statistics()              # sharded job: one small all-reduce per rollout (no-op on one GPU)
        if self._collect_fused and self._close_gae:
            self.adv, self.ret = self._adv_buf, self._ret_buf      # written by fw_collect_close inside the rollout
        else:
            self.adv, self.ret = self._gae(self.buf_rew, self.buf_val, self.buf_start, self.last_values, self.last_starts,
                                           cfg.gamma, cfg.gae_lambda)
        td = _dist()
        steps = cfg.n_steps * env.num_envs * (td.get_world_size() if td is not None else 1)
        self._void_steps += steps                  # (since the last clean look at the status word)
        self.num_timesteps += steps

    # ---- SB3 PPO.train ------------------------------------------------------------------------
    def _minibatch_step(self, obs, act, old_logp, adv, ret, idx, g_mean, g_std, params):
        cfg = self.cfg
        kw = {"img": self.buf_img.reshape((-1,) + tuple(self.buf_img.shape[2:]))[idx]} if self._img else {}
        a = adv[idx]
        if cfg.normalize_advantage and a.numel() > 1:
            a = (a - g_mean) / (g_std + 1e-8) if cfg.adv_norm_scope == "global" else (a - a.mean()) / (a.std() + 1e-8)
        values, logp, entropy = self.policy.evaluate_actions(obs[idx], act[idx], **kw)
        ratio = torch.exp(logp - old_logp[idx])
        policy_loss = -torch.min(a * ratio, a * ratio.clamp(1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
        value_loss = torch.nn.functional.mse_loss(ret[idx], values)
        entropy_loss = -entropy.mean()
        loss = policy_loss + cfg.ent_coef * entropy_loss + cfg.vf_coef * value_loss
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        if not self._replicated:
            allreduce_grads_(params)           # "allreduce" mode only: a replicated update needs no collective here
        torch.nn.utils.clip_grad_norm_(params, cfg.max_grad_norm)
        self.optimizer.step()
        self._loss_acc += torch.stack([policy_loss.detach(), value_loss.detach(), entropy_loss.detach()])

    def _update_buffers(self):
        """``(obs, act, old_logp, adv, ret)`` flattened to [B, ...] as the update walks them: the rank's own rollout, or --
        replicated update of a sharded job -- the rollouts of ALL ranks, all-gathered once (obs / actions / log-probs /
        advantages / returns packed into one [T, N, D + 7] float32 tensor: one collective of ~9 MB per 65 536 samples) and
        laid out exactly like the buffer of one big job over the concatenated envs ([T, W * N], rank-major env order)."""
        T, N, D = self.cfg.n_steps, self.env.num_envs, self.env.obs_dim
        if not self._replicated:
            B = T * N
            return (self.buf_obs.reshape(B, -1), self.buf_act.reshape(B, -1), self.buf_logp.reshape(B),
                    self.adv.reshape(B), self.ret.reshape(B))
        pack = torch.cat([self.buf_obs, self.buf_act, self.buf_logp.unsqueeze(-1), self.adv.unsqueeze(-1), self.ret.unsqueeze(-1)], dim=-1)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        g = all_gather_cat(pack, dim=1)                          # [T, W * N, D + 7]
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.allgather_ms = (time.perf_counter() - t0) * 1e3
        self.allgather_bytes = pack.numel() * 4
        B = g.shape[0] * g.shape[1]
        g = g.reshape(B, D + 7)
        if self._gathered is None:                               # persistent: a captured update graph holds these addresses
            f32 = dict(dtype=torch.float32, device=self.device)
            self._gathered = (torch.empty((B, D), **f32), torch.empty((B, 4), **f32), torch.empty(B, **f32),
                              torch.empty(B, **f32), torch.empty(B, **f32))
        o, a, lp, ad, rt = self._gathered
        o.copy_(g[:, :D]); a.copy_(g[:, D:D + 4]); lp.copy_(g[:, D + 4]); ad.copy_(g[:, D + 5]); rt.copy_(g[:, D + 6])
        return self._gathered

    def replica_checksum(self) -> float:
        """Sum over ranks of |own parameters - rank 0's parameters|: exactly 0.0 while the replicas are bit-identical."""
        flat = torch.cat([p.detach().reshape(-1) for p in self.policy.parameters()]).to(torch.float64)
        td = _dist()
        if td is None:
            return 0.0
        ref = all_gather_cat(flat.unsqueeze(0), dim=0)[0]
        return float(all_reduce_sum_((flat - ref).abs().sum().reshape(1)).item())

    def train(self):
        cfg = self.cfg
        st, st_all = self._collect_status_word(wait=True, collective=True)     # a void rollout must not reach the update -- on any rank
        if st_all:
            msg = self._take_back_void_rollout(st, st_all)
            if not self._void_recoverable:
                raise RuntimeError(msg)
            self.collect_rollouts()                    # the same rollout again, on the collector without in-grid waits (every rank: st_all is agreed)
        if getattr(self, "adv", None) is None:
            raise RuntimeError("train() without a rollout: the last one was void (check_collect_status raised) -- call collect_rollouts() first")
        obs, act, old_logp, adv, ret = self._update_buffers()
        B = obs.shape[0]
        if self._replicated:                                     # the gathered advantages ARE the global ones
            g_mean, g_std = adv.mean(), adv.std()
        else:
            g_mean, g_std, _ = global_advantage_stats(adv)       # "allreduce" mode: all-gather of the advantages only
        params = list(self.policy.parameters())
        self._loss_acc.zero_()          # persistent buffer: the captured update graph holds its address
        nb = 0
        bs = cfg.batch_size
        # the fused kernel walks whole minibatches of the buffer it is handed and has no gradient exchange: a multi-process job
        # may only take it on the gathered buffer ("replicated"); dist_update="allreduce" stays on the torch path below, whose
        # minibatch loop all-reduces the gradients (allreduce_grads_)
        fused_ok = _dist() is None or self._replicated
        if fused_ok and B % bs == 0 and FusedPpoUpdate.applies(self.policy, cfg, self.env.obs_dim, bs, self.device):
            # the whole minibatch sequence in one kernel; the permutations are drawn exactly like the loop below
            if self._fused is None:
                self._fused = FusedPpoUpdate(self.policy, self.optimizer, self.env.obs_dim)
            perm = torch.cat([torch.randperm(B, device=self.device, generator=self.perm_gen) for _ in range(cfg.n_epochs)]).to(torch.int32)
            nb = cfg.n_epochs * (B // bs)
            err = None
            try:
                la = self._fused.run(cfg, obs.contiguous(), act.contiguous(), old_logp.contiguous(), adv.contiguous(), ret.contiguous(),
                                     perm, nb, float(g_mean), float(g_std))
            except RuntimeError as e:          # a workgroup of the launch gave up (status word): nothing was written back on this rank
                err = e
            if self._replicated:               # ... and in a sharded job no rank may go on alone: the replicas would part ways
                bad = all_reduce_sum_(torch.tensor([1.0 if err is not None else 0.0], dtype=torch.float64, device=self.device))
                if float(bad.item()) > 0 and err is None:
                    err = RuntimeError("fw_ppo_update gave up on another rank of the job; this rank's update is discarded with it")
            if err is not None:
                self._flat_current = False     # the flat image holds a result that was not committed
                self._fused.synced = False
                raise err
            self._fused.commit()
            self._g_update = None              # the torch-path graph (if any) holds stale Adam state
            self._flat_current = True          # store_to_torch left flat == the module parameters
            self.logs = {"policy_loss": la[0], "value_loss": la[1], "entropy_loss": la[2],
                         "adv_mean": float(g_mean), "adv_std": float(g_std)}
            return
        self._flat_current = False             # the torch path below moves the module parameters
        if self._fused is not None:
            self._fused.synced = False
        use_graph = self._graphs and B % bs == 0
        if use_graph and self._g_update is None:
            self._idx = torch.zeros(bs, dtype=torch.long, device=self.device)
            self._gm = torch.zeros((), device=self.device); self._gs = torch.ones((), device=self.device)
            self._adv_s, self._ret_s = torch.zeros(B, device=self.device), torch.zeros(B, device=self.device)
            self._adv_s.copy_(adv); self._ret_s.copy_(ret); self._gm.copy_(g_mean); self._gs.copy_(g_std)
            self._idx.copy_(torch.arange(bs, device=self.device))
            sd_p = [p.detach().clone() for p in params]
            sd_o = {p: {k: v.clone() for k, v in stt.items() if torch.is_tensor(v)} for p, stt in self.optimizer.state.items()}
            st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):                               # warm-up on the side stream, then undone below
                    self._minibatch_step(obs, act, old_logp, self._adv_s, self._ret_s, self._idx, self._gm, self._gs, params)
            torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
            self._g_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_update):
                self._minibatch_step(obs, act, old_logp, self._adv_s, self._ret_s, self._idx, self._gm, self._gs, params)
            with torch.no_grad():                                # undo the warm-up: weights and Adam moments / step
                for p, q in zip(params, sd_p):
                    p.copy_(q)
                for p, stt in self.optimizer.state.items():
                    for k, v in stt.items():
                        if torch.is_tensor(v):
                            if p in sd_o and k in sd_o[p]:
                                v.copy_(sd_o[p][k])
                            else:
                                v.zero_()
            self._loss_acc.zero_()
        if use_graph:
            self._adv_s.copy_(adv); self._ret_s.copy_(ret); self._gm.copy_(g_mean); self._gs.copy_(g_std)
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(B, device=self.device, generator=self.perm_gen)
            for s in range(0, B, bs):
                if use_graph:
                    self._idx.copy_(perm[s:s + bs])
                    self._g_update.replay()
                else:
                    self._minibatch_step(obs, act, old_logp, adv, ret, perm[s:s + bs], g_mean, g_std, params)
                nb += 1
        la = (self._loss_acc / nb).tolist()                      # the only host sync of the update
        self.logs = {"policy_loss": la[0], "value_loss": la[1], "entropy_loss": la[2],
                     "adv_mean": float(g_mean), "adv_std": float(g_std)}

    # does self._fused.flat hold the current policy parameters?  Clearing it also tells the fused update that its images (parameters
    # AND Adam moments) may be behind the module / optimiser: it then reloads them at its next call instead of reusing them
    @property
    def _flat_current(self) -> bool:
        return self.__dict__.get("_flat_current_", False)

    @_flat_current.setter
    def _flat_current(self, v: bool) -> None:
        self.__dict__["_flat_current_"] = bool(v)
        f = self.__dict__.get("_fused")
        if not v and f is not None:
            f.synced = False

    @property
    def world_size(self) -> int:
        td = _dist()
        return td.get_world_size() if td is not None else 1

    @property
    def rank(self) -> int:
        td = _dist()
        return td.get_rank() if td is not None else 0

    def learn(self, total_timesteps: int, callbacks=(), reset_num_timesteps: bool = True):
        """``model.learn(total_timesteps, reset_num_timesteps, callback=[eval, checkpoint])``
        (train/train_Fixedwing_Waypoints_v3.py:329-334).  Callbacks get ``on_rollout_end(ppo)``
        after every update; returning False stops training."""
        if reset_num_timesteps:
            self.num_timesteps = 0
        else:
            total_timesteps += self.num_timesteps          # SB3: continue for another total_timesteps
        go = True
        while go and self.num_timesteps < total_timesteps:
            self.collect_rollouts()
            self.train()
            for cb in callbacks:
                go = bool(cb.on_rollout_end(self)) and go
        for cb in callbacks:                               # (an evaluation that runs beside the training is collected here)
            if hasattr(cb, "on_training_end"):
                cb.on_training_end(self)
        return self

    # ---- checkpoint (model + normaliser), train/train_Fixedwing_Waypoints_v3.py:340-347 ----------
    def state_dict(self):
        self.check_collect_status()                # never checkpoint behind a void rollout
        return {"policy": self.policy.state_dict(), "optimizer": self.optimizer.state_dict(),
                "vecnormalize": self.env.state_dict(), "num_timesteps": self.num_timesteps}

    def invalidate_graphs(self) -> None:
        """Drop the captured rollout / update graphs (their kernel arguments froze scalars and state addresses that a
        checkpoint load may have replaced); the next rollout runs eagerly once and re-captures."""
        self._g_rollout = self._g_update = None
        self._warm_rollouts = 0

    def load_state_dict(self, sd, reset_num_timesteps: bool = True):
        self.policy.load_state_dict(sd["policy"]); self.optimizer.load_state_dict(sd["optimizer"])
        self.invalidate_graphs()           # the optimiser's state tensors were replaced, normaliser scalars may differ
        self._flat_current = False
        if self._fused is not None:
            self._fused.synced = False
        self.env.load_state_dict(sd["vecnormalize"])
        self.num_timesteps = 0 if reset_num_timesteps else int(sd["num_timesteps"])
