"""Checkpoint format: policy + optimiser + normaliser statistics + env state.

The reference's only persistence path is SB3's ``model.save`` (``final_model.zip``,
``best_model.zip``, ``waypoints_ppo_<steps>_steps.zip`` from ``CheckpointCallback``) plus
``VecNormalize.save`` (``vecnorm.pkl``)
-- train/train_Fixedwing_Waypoints_v3.py:64-80 (vecnorm path inference), :254-258 (load the
normaliser, keep training it), :281-285 (checkpoint callback), :313-327 (load *parameters
only* into a freshly configured PPO and **reset** the timestep counter), :340-347 (final save).

One ``.pt`` file (``torch.save`` of tensors and plain Python scalars only, so it loads with
``weights_only=True``) holds what those files hold, and additionally the simulator state
(``fw_get_state``: ``float64[N, FW_STATE_DIM]`` + seed), which the reference cannot save
because a Bullet world is not serialisable -- a resumed run here can continue the very
episodes it was in.

    save(path, ppo)                     full checkpoint
    load(path, ppo, reset_num_timesteps=True, restore_env_state=False)
    set_parameters(path, ppo)           policy parameters only (reference :313-320)
    save_vecnormalize / load_vecnormalize   the ``vecnorm.pkl`` twin
    infer_vecnorm_path                  reference :64-80
    CheckpointCallback                  periodic ``<prefix>_<num_timesteps>_steps.pt``
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import tempfile
from typing import Optional

import numpy as np
import torch

from . import config as K

FORMAT_VERSION = 1


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_cpu(v) for v in obj]
    return obj


def config_fingerprint(cfg) -> str:
    """sha256 of the fw_config bytes: the saved simulator state only means the same thing under the same config."""
    return hashlib.sha256(C.string_at(C.addressof(cfg), C.sizeof(cfg))).hexdigest()


def _atomic_torch_save(obj, path: str) -> None:
    """Write through a uniquely named temp file in the target directory, then rename: concurrent writers (several
    ranks, several jobs) can never interleave into one file or remove each other's temp file."""
    d = os.path.dirname(os.path.abspath(path)) or "."
    os.makedirs(d, exist_ok=True)
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=d)
    try:
        with os.fdopen(fd, "wb") as f:
            torch.save(obj, f)
        os.replace(tmp, path)                 # never leave a half-written checkpoint behind
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


def save(path: str, ppo, include_env_state: bool = True) -> str:
    _atomic_torch_save(snapshot(ppo, include_env_state), path)
    return path


def snapshot(ppo, include_env_state: bool = True) -> dict:
    """What :func:`save` writes, as host tensors: taken now, written later (:func:`write`) -- the evaluation callback keeps
    the weights it is evaluating until it knows whether they are the best so far."""
    if hasattr(ppo, "check_collect_status"):
        ppo.check_collect_status()             # never snapshot behind a void rollout (as PPO.state_dict(); the best-model path comes through here)
    venv = ppo.env.venv
    sd = {
        "format_version": FORMAT_VERSION,
        "policy": _to_cpu(ppo.policy.state_dict()),
        "optimizer": _to_cpu(ppo.optimizer.state_dict()),
        "vecnormalize": _to_cpu(ppo.env.state_dict()),
        "num_timesteps": int(ppo.num_timesteps),
        "obs_dim": int(ppo.env.obs_dim),
        "num_envs": int(ppo.env.num_envs),
        "abi_version": int(K.FW_ABI_VERSION),
        "state_dim": int(K.FW_STATE_DIM),
    }
    if include_env_state and hasattr(venv, "get_state"):
        sd["env_state"] = torch.from_numpy(np.ascontiguousarray(venv.get_state()))
        # the episodes that follow are a function of (seed, global env id, episode): record what the state was drawn under
        sd["env_seed"] = int(getattr(venv, "seed_value", 0))
        sd["global_env_offset"] = int(getattr(venv, "global_env_offset", 0))
        if hasattr(venv, "cfg"):
            sd["config_sha256"] = config_fingerprint(venv.cfg)
        sd["env_returns"] = ppo.env.returns.detach().cpu()
        if hasattr(venv, "obs") and torch.is_tensor(venv.obs):
            sd["env_obs"] = venv.obs.detach().cpu()               # the raw observation the next rollout's first step reads
        if getattr(ppo, "last_obs", None) is not None:
            sd["last_obs"] = ppo.last_obs.detach().cpu()
            sd["last_starts"] = ppo.last_starts.detach().cpu()
        sd["sampler_rng_state"] = ppo.gen.get_state().cpu()        # action-sampling generator: resume draws the same actions
        if getattr(ppo, "_collect_fused", False):
            sd["collect_rng"] = ppo._rng.detach().cpu()             # (seed, draw counter) of fw_policy_act
        if getattr(ppo, "perm_gen", None) is not None and ppo.perm_gen is not ppo.gen:
            sd["perm_rng_state"] = ppo.perm_gen.get_state().cpu()   # replicated update: the shared minibatch-permutation stream
    return sd


def write(sd: dict, path: str) -> str:
    _atomic_torch_save(sd, path)
    return path


def _read(path: str) -> dict:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if sd.get("format_version") != FORMAT_VERSION:
        raise ValueError(f"{path}: unsupported checkpoint format_version {sd.get('format_version')!r}")
    return sd


def load(path: str, ppo, reset_num_timesteps: bool = True, restore_env_state: bool = False) -> dict:
    """Restore policy, optimiser and normaliser.  ``reset_num_timesteps=True`` is what the
    reference does on every (re)start (:322-324); ``False`` continues the counter.
    ``restore_env_state=True`` additionally puts every env back into the saved simulator state
    (same ``num_envs`` and ABI required) so the interrupted episodes continue."""
    sd = _read(path)
    if sd["obs_dim"] != ppo.env.obs_dim:
        raise ValueError(f"checkpoint obs_dim {sd['obs_dim']} != env obs_dim {ppo.env.obs_dim}")
    ppo.load_state_dict(sd, reset_num_timesteps=reset_num_timesteps)
    if restore_env_state:
        if "env_state" not in sd:
            raise ValueError(f"{path} holds no env state")
        if sd["num_envs"] != ppo.env.num_envs or sd["state_dim"] != K.FW_STATE_DIM or sd["abi_version"] != K.FW_ABI_VERSION:
            raise ValueError("env state in the checkpoint does not fit this env (num_envs / FW_STATE_DIM / ABI version)")
        venv = ppo.env.venv
        for key, have in (("env_seed", int(getattr(venv, "seed_value", 0))), ("global_env_offset", int(getattr(venv, "global_env_offset", 0))),
                          ("config_sha256", config_fingerprint(venv.cfg) if hasattr(venv, "cfg") else None)):
            if key in sd and have is not None and sd[key] != have:
                raise ValueError(f"env state in the checkpoint was saved under a different {key} ({sd[key]!r} != {have!r}): "
                                 "the episodes that follow would not be those of the interrupted run")
        ppo.env.venv.set_state(sd["env_state"].numpy())
        ppo.env.returns.copy_(sd["env_returns"].to(ppo.env.returns.device))
        if "env_obs" in sd and hasattr(venv, "obs"):
            venv.obs.copy_(sd["env_obs"].to(venv.obs.device))
        elif hasattr(venv, "observe_tensor"):
            venv.observe_tensor()          # older checkpoint: recompute the raw observation of the restored state (fw_observe)
        if "last_obs" in sd:
            if ppo.last_obs is None:
                ppo.last_obs = sd["last_obs"].to(ppo.device).clone()
            else:
                ppo.last_obs.copy_(sd["last_obs"].to(ppo.device))        # in place: a captured rollout graph holds its address
            ppo.last_starts.copy_(sd["last_starts"].to(ppo.device))
            if not hasattr(ppo, "last_values"):
                ppo.last_values = torch.zeros(ppo.env.num_envs, dtype=torch.float32, device=ppo.device)
        if "sampler_rng_state" in sd:
            ppo.gen.set_state(sd["sampler_rng_state"])
        if "collect_rng" in sd and getattr(ppo, "_collect_fused", False):
            ppo._rng.copy_(sd["collect_rng"].to(ppo.device))
        if "perm_rng_state" in sd and getattr(ppo, "perm_gen", None) is not None and ppo.perm_gen is not ppo.gen:
            ppo.perm_gen.set_state(sd["perm_rng_state"])
    return sd


def set_parameters(path: str, ppo) -> None:
    """``model.set_parameters(pretrained.get_parameters())``: weights (and Adam moments) only;
    hyper-parameters, normaliser and counters stay those of the freshly configured ``ppo``."""
    sd = _read(path)
    ppo.policy.load_state_dict(sd["policy"])
    ppo.optimizer.load_state_dict(sd["optimizer"])
    ppo._g_update = None                     # new optimiser state tensors: the captured update graph is stale
    ppo._flat_current = False
    ppo.invalidate_graphs()


def save_vecnormalize(path: str, env) -> str:
    _atomic_torch_save({"format_version": FORMAT_VERSION, "vecnormalize": _to_cpu(env.state_dict())}, path)
    return path


def load_vecnormalize(path: str, env, training: Optional[bool] = None, norm_reward: Optional[bool] = None):
    """``VecNormalize.load(path, venv)`` followed by the flag overrides the reference applies
    (:256-258 train: training=True, norm_reward=True; :266-268 eval: False, False)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    env.load_state_dict(sd["vecnormalize"])
    if training is not None:
        env.training = training
    if norm_reward is not None:
        env.norm_reward = norm_reward
    env.version += 1          # scalars frozen into a captured rollout graph (clip, gamma, flags) may have changed
    return env


def infer_vecnorm_path(pretrained_model: Optional[str], vecnorm_path: Optional[str], model_dir: Optional[str] = None,
                       name: str = "vecnorm.pt") -> Optional[str]:
    """train/train_Fixedwing_Waypoints_v3.py:64-80: explicit path, else next to the model, else model_dir."""
    if vecnorm_path:
        return vecnorm_path
    if not pretrained_model:
        return None
    for d in (os.path.dirname(pretrained_model), model_dir):
        if d:
            p = os.path.join(d, name)
            if os.path.exists(p):
                return p
    return None


class CheckpointCallback:
    """SB3 ``CheckpointCallback(save_freq, save_path, name_prefix)`` (:281-285); ``save_freq``
    counts vec-steps, the file is named after the global timestep count."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rl_model", include_env_state: bool = False):
        self.save_freq, self.save_path, self.name_prefix = max(int(save_freq), 1), save_path, name_prefix
        self.include_env_state = include_env_state
        self._next = self.save_freq
        self.saved = []

    def on_rollout_end(self, ppo) -> bool:
        n_calls = ppo.num_timesteps // max(ppo.env.num_envs * ppo.world_size, 1)
        if n_calls >= self._next:
            while self._next <= n_calls:
                self._next += self.save_freq
            if ppo.rank == 0:
                p = os.path.join(self.save_path, f"{self.name_prefix}_{ppo.num_timesteps}_steps.pt")
                self.saved.append(save(p, ppo, include_env_state=self.include_env_state))
        return True
