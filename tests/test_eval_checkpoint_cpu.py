"""Eval harness + checkpoint format on a CPU stand-in env:
host logic only -- episode accounting like SB3's evaluate_policy, the reference's eval scalars,
normaliser sync, file round trips, counter reset / continue."""
import os

import numpy as np
import pytest
import torch

import pyflyt_drone_amd as P
from pyflyt_drone_amd import checkpoint, evaluate
from pyflyt_drone_amd import config as K
from pyflyt_drone_amd.rollout import PPO, MlpPolicy, PPOConfig, VecNormalizeDevice, gae_reference
from test_rollout_cpu import FakeVenv


class FakeTaskVenv(FakeVenv):
    """FakeVenv + the `rewards` / `info` attributes and get/set_state of the device env."""

    def __init__(self, n=6, d=6, seed=0, horizon=5):
        super().__init__(n, d, seed, horizon)
        self.rewards = torch.zeros(n, dtype=torch.float64)
        self.info = torch.zeros((n, K.FW_INFO_DIM), dtype=torch.int32)
        self.episodes = torch.zeros(n, dtype=torch.long)

    def step_tensor(self, actions):
        obs, rew, term, trunc = super().step_tensor(actions)
        self.rewards = rew
        done = (term | trunc).bool()
        self.episodes += done.long()
        # deterministic per-env "targets reached": env i reaches i % 4 targets in every episode
        self.info[:, K.INFO_NUM_TARGETS_REACHED] = (torch.arange(self.num_envs) % 4).to(torch.int32)
        self.info[:, K.INFO_ENV_COMPLETE] = ((torch.arange(self.num_envs) % 4) == 3).to(torch.int32)
        return obs, rew, term, trunc

    def get_state(self):
        s = np.zeros((self.num_envs, K.FW_STATE_DIM)); s[:, 0] = self.t.numpy(); s[:, 1] = self.episodes.numpy()
        return s

    def set_state(self, s):
        self.t = torch.as_tensor(s[:, 0]).long(); self.episodes = torch.as_tensor(s[:, 1]).long()


def _eval_env(n=6):
    return VecNormalizeDevice(FakeTaskVenv(n), training=False, norm_reward=False, use_fused_kernel=False)


def test_evaluate_policy_counts_episodes_like_sb3_and_reports_reference_scalars():
    env = _eval_env(6)
    pol = MlpPolicy(env.obs_dim)
    seen = []
    r = evaluate.evaluate_policy(pol, env, n_eval_episodes=15, deterministic=True, callback=seen.append)
    assert len(r.episode_rewards) == len(r.episode_lengths) == 15 == len(seen)
    # SB3 targets: (n_eval_episodes + i) // n_envs episodes from env i  -> 2,2,2,3,3,3
    per_env = np.bincount([i["num_targets_reached"] for i in seen], minlength=4)        # env i reports i % 4
    assert per_env.tolist() == [2 + 3, 2 + 3, 2, 3]                                     # envs {0,4},{1,5},{2},{3}
    assert all(1 <= l <= 7 for l in r.episode_lengths) and max(r.episode_lengths) >= 5          # horizon 5 + (i % 3), or an early termination
    sc = r.scalars(num_targets_total=3)
    reached = np.array(r.num_targets_reached)
    for i in (1, 2, 3):
        assert sc[f"eval/wp{i}_reach_rate"] == pytest.approx(np.mean(reached >= i))
    assert sc["eval/success_rate"] == pytest.approx(np.mean(reached == 3))
    assert sc["eval/mean_reward"] == pytest.approx(np.mean(r.episode_rewards))
    assert "eval/duck_strike_rate" not in sc
    # un-normalised rewards: the eval wrapper must not touch its statistics
    assert float(env.obs_rms.count) == pytest.approx(1e-4)


def test_sync_envs_normalization_copies_statistics():
    tr = VecNormalizeDevice(FakeTaskVenv(6), use_fused_kernel=False)
    ev = _eval_env(6)
    tr.reset()
    for _ in range(5):
        tr.step(torch.zeros((6, 4), dtype=torch.float64))
    evaluate.sync_envs_normalization(tr, ev)
    assert torch.equal(tr.obs_rms.mean, ev.obs_rms.mean) and torch.equal(tr.obs_rms.var, ev.obs_rms.var)
    assert ev.obs_rms.mean.data_ptr() != tr.obs_rms.mean.data_ptr()
    x = torch.randn(6, 6, dtype=torch.float64)
    assert torch.equal(tr.normalize_obs(x), ev.normalize_obs(x))


def _ppo(seed=3, n=8):
    env = VecNormalizeDevice(FakeTaskVenv(n, seed=seed), use_fused_kernel=False)
    return PPO(env, PPOConfig(n_steps=8, batch_size=16, n_epochs=2, seed=seed, use_graphs=False), gae_fn=gae_reference)


def test_checkpoint_round_trip_resumes_bit_identically(tmp_path):
    a = _ppo()
    a.learn(2 * 8 * 8)
    path = checkpoint.save(str(tmp_path / "ck" / "model.pt"), a)
    assert os.path.exists(path) and not os.path.exists(path + ".tmp")
    gen_state = a.gen.get_state()
    venv_gen = a.env.venv.g.get_state()
    a.learn(8 * 8, reset_num_timesteps=False)
    ref = [p.detach().clone() for p in a.policy.parameters()]

    b = _ppo()
    sd = checkpoint.load(path, b, reset_num_timesteps=False, restore_env_state=True)
    assert b.num_timesteps == sd["num_timesteps"] == 128
    b.gen.set_state(gen_state); b.env.venv.g.set_state(venv_gen)      # (the fake env's own RNG is not part of the format)
    b.learn(8 * 8, reset_num_timesteps=False)
    assert b.num_timesteps == 192
    for p, q in zip(ref, b.policy.parameters()):
        assert torch.equal(p, q)
    assert torch.equal(a.env.obs_rms.mean, b.env.obs_rms.mean) and float(a.env.ret_rms.var) == float(b.env.ret_rms.var)


def test_reference_restart_semantics_parameters_only_and_counter_reset(tmp_path):
    a = _ppo()
    a.learn(128)
    path = checkpoint.save(str(tmp_path / "final_model.pt"), a, include_env_state=False)
    vn = checkpoint.save_vecnormalize(str(tmp_path / "vecnorm.pt"), a.env)
    assert checkpoint.infer_vecnorm_path(path, None) == vn                       # next to the model (reference :64-80)
    assert checkpoint.infer_vecnorm_path(path, "/x/y.pt") == "/x/y.pt"
    assert checkpoint.infer_vecnorm_path(None, None) is None
    b = _ppo(seed=11)
    checkpoint.set_parameters(path, b)                                           # model.set_parameters(...)
    for p, q in zip(a.policy.parameters(), b.policy.parameters()):
        assert torch.equal(p, q)
    assert b.num_timesteps == 0 and float(b.env.obs_rms.count) == pytest.approx(1e-4)   # counters / normaliser untouched
    checkpoint.load_vecnormalize(vn, b.env, training=True, norm_reward=True)
    assert torch.equal(a.env.obs_rms.mean, b.env.obs_rms.mean) and b.env.training and b.env.norm_reward
    ev = _eval_env(8)
    checkpoint.load_vecnormalize(vn, ev, training=False, norm_reward=False)      # eval twin (reference :266-268)
    assert not ev.training and not ev.norm_reward
    c = _ppo(seed=12)
    checkpoint.load(path, c)                                                     # default: counter reset (reference :322-324)
    assert c.num_timesteps == 0
    with pytest.raises(ValueError):
        checkpoint.load(path, c, restore_env_state=True)                         # no env state in this file
    bad = VecNormalizeDevice(FakeTaskVenv(8, d=7), use_fused_kernel=False)
    with pytest.raises(ValueError):
        checkpoint.load(path, PPO(bad, PPOConfig(n_steps=8, batch_size=16, n_epochs=1, use_graphs=False), gae_fn=gae_reference))


def test_eval_and_checkpoint_callbacks_fire_on_the_reference_cadence(tmp_path):
    a = _ppo()
    ev = evaluate.EvalCallback(_eval_env(4), n_eval_episodes=6, eval_freq=16, log_path=str(tmp_path / "logs"),
                               best_model_save_path=str(tmp_path / "models"), num_targets_total=3)
    ck = checkpoint.CheckpointCallback(save_freq=24, save_path=str(tmp_path / "models"), name_prefix="waypoints_ppo")
    a.learn(8 * 8 * 6, callbacks=[ev, ck])                     # 6 rollouts x 8 vec-steps = 48 vec-steps
    assert ev.n_evals == 3 and ev.evaluations_timesteps == [128, 256, 384]          # at 16, 32, 48 vec-steps
    assert [os.path.basename(p) for p in ck.saved] == ["waypoints_ppo_192_steps.pt", "waypoints_ppo_384_steps.pt"]
    z = np.load(str(tmp_path / "logs" / "evaluations.npz"), allow_pickle=True)
    assert z["timesteps"].tolist() == [128, 256, 384] and len(z["results"]) == 3 and len(z["results"][0]) == 6
    assert os.path.exists(tmp_path / "models" / "best_model.pt")
    assert ev.last_scalars["time/total_timesteps"] == 384 and "eval/wp3_reach_rate" in ev.last_scalars
    assert ev.best_mean_reward >= ev.last_mean_reward
