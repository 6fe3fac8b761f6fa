"""Eval harness and checkpoint/resume on the real device envs."""
import os

import numpy as np
import pytest
import torch

import pyflyt_drone_amd as P
from pyflyt_drone_amd import checkpoint, evaluate
from pyflyt_drone_amd import config as K
from pyflyt_drone_amd import rollout as R

pytestmark = pytest.mark.gpu


def _wp_cfg():
    # short episodes: low start, small dome -> crashes / OOB within a few dozen steps
    return K.train_waypoints_v3_config(flight_dome_size=30.0, max_duration_seconds=4.0)


def test_eval_harness_waypoints_metrics_are_consistent_with_the_env_info():
    venv = P.FixedwingVecEnv(_wp_cfg(), 64, seed=5)
    env = R.VecNormalizeDevice(venv, training=False, norm_reward=False)
    pol = R.MlpPolicy(env.obs_dim).cuda()
    infos = []
    r = evaluate.evaluate_policy(pol, env, n_eval_episodes=100, deterministic=True, callback=infos.append)
    assert len(r.episode_rewards) == 100 == len(infos)
    assert float(env.obs_rms.count) == pytest.approx(1e-4)            # frozen statistics
    for i in infos:                                                   # every episode ended for a reason the env reported
        assert i["episode"]["l"] >= 1
        assert i["collision"] or i["out_of_bounds"] or i["env_complete"] or i["episode"]["l"] >= K.max_steps(venv.cfg)
        assert i["is_success"] == i["env_complete"]
        # reference reward bounds: -0.1 per step, -100 on crash/OOB, +100 per waypoint (sparse)
        assert i["episode"]["r"] <= 100.0 * i["num_targets_reached"] + 1e-9
    sc = r.scalars(num_targets_total=8)
    rates = [sc[f"eval/wp{k}_reach_rate"] for k in range(1, 9)]
    assert all(a >= b for a, b in zip(rates, rates[1:]))              # reaching k+1 implies reaching k
    assert sc["eval/success_rate"] == pytest.approx(rates[-1])
    # a second evaluation of the same deterministic policy on a re-seeded env reproduces the first
    venv.seed(5)
    r2 = evaluate.evaluate_policy(pol, env, n_eval_episodes=100, deterministic=True)
    assert sorted(r2.episode_lengths) == sorted(r.episode_lengths)


@pytest.mark.parametrize("task", ["waypoints", "objlock"])
def test_replayed_evaluation_returns_the_episodes_of_the_step_by_step_loop(task):
    """evaluate_policy's default on the GPU (hipGraph replays of 8 vec-steps, bookkeeping on the device, one look at the
    counters per replay) against the plain loop that reads back after every step: same episodes, same order, same info,
    including envs that finish several episodes and the uneven split of n_eval_episodes over the envs."""
    cfg = _wp_cfg() if task == "waypoints" else K.train_objlock_config(max_duration_seconds=3.0)
    pol = None
    out = []
    for use_graph in (False, True):
        venv = P.FixedwingVecEnv(cfg, 24, seed=9)
        env = R.VecNormalizeDevice(venv, training=False, norm_reward=False)
        if pol is None:
            torch.manual_seed(0)
            pol = R.MlpPolicy(env.obs_dim).cuda()
        infos = []
        r = evaluate.evaluate_policy(pol, env, n_eval_episodes=61, deterministic=True, callback=infos.append, use_graph=use_graph,
                                     use_fused=False)             # (the torch ops on both sides: the same bits)
        out.append((r, infos))
    (a, ia), (b, ib) = out
    assert len(a.episode_rewards) == 61 == len(b.episode_rewards)
    assert a.episode_lengths == b.episode_lengths and a.episode_rewards == b.episode_rewards
    assert a.num_targets_reached == b.num_targets_reached and a.is_success == b.is_success and a.duck_strike == b.duck_strike
    assert ia == ib


@pytest.mark.parametrize("task", ["waypoints", "objlock"])
def test_fused_evaluation_flies_the_episodes_of_the_torch_evaluation(task):
    """evaluate_policy's default where it applies: a vec-step of the evaluation is ONE fw_collect_step launch (deterministic, statistics
    frozen) instead of the framework ops.  The policy forward is then the kernel's (fp32 MFMA), equal to torch's to rounding: the
    episodes are the same ones -- same number, same order, lengths and outcomes equal but for an episode that ends on a knife's edge,
    rewards equal to ~1e-4 relative."""
    cfg = _wp_cfg() if task == "waypoints" else K.train_objlock_config(max_duration_seconds=3.0)
    pol = None
    out = []
    for fused in (False, True):
        venv = P.FixedwingVecEnv(cfg, 24, seed=9)
        env = R.VecNormalizeDevice(venv, training=False, norm_reward=False)
        with torch.no_grad():                                      # statistics as after some training: not the identity
            env.obs_rms.mean.copy_(torch.linspace(-0.2, 0.3, env.obs_dim, dtype=torch.float64, device="cuda"))
            env.obs_rms.var.copy_(torch.linspace(0.5, 2.0, env.obs_dim, dtype=torch.float64, device="cuda"))
        if pol is None:
            torch.manual_seed(0)
            pol = R.MlpPolicy(env.obs_dim).cuda()
        infos = []
        job = evaluate.ReplayedEvaluation(pol, env, __import__("numpy").array([(61 + i) // 24 for i in range(24)]), infos.append, use_fused=fused)
        assert job.fused == fused
        out.append((job.run(None), infos))
    (a, ia), (b, ib) = out
    assert len(a.episode_rewards) == 61 == len(b.episode_rewards)
    same = [x == y for x, y in zip(a.episode_lengths, b.episode_lengths)]
    assert sum(same) >= 58, (a.episode_lengths, b.episode_lengths)
    for k, ok in enumerate(same):
        if ok:
            assert b.episode_rewards[k] == pytest.approx(a.episode_rewards[k], rel=2e-3, abs=2e-3), k
    assert sum(x == y for x, y in zip(a.is_success, b.is_success)) >= 58


def test_fw_eval_track_is_the_bookkeeping_of_the_torch_loop():
    """fw_eval_track (one launch) against the framework ops it replaces in evaluate.ReplayedEvaluation._body, on random step outputs:
    accumulators, slots of finished episodes, info rows, episode counts and the step counter agree exactly over 200 steps."""
    import ctypes as C
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    n, E, idim = 37, 3, K.FW_INFO_DIM
    tg = torch.randint(1, E + 1, (n,), device="cuda", generator=g)
    def fresh():
        return dict(counts=torch.zeros(n, dtype=torch.int64, device="cuda"), cur_rew=torch.zeros(n, dtype=torch.float64, device="cuda"),
                    cur_len=torch.zeros(n, dtype=torch.int64, device="cuda"), step=torch.zeros((), dtype=torch.int64, device="cuda"),
                    fin_rew=torch.zeros((n, E), dtype=torch.float64, device="cuda"), fin_len=torch.zeros((n, E), dtype=torch.int64, device="cuda"),
                    fin_step=torch.zeros((n, E), dtype=torch.int64, device="cuda"), fin_info=torch.zeros((n, E, idim), dtype=torch.int32, device="cuda"))
    a, b = fresh(), fresh()
    ar = torch.arange(n, device="cuda")
    for dtype in (torch.float64, torch.float32):
        for t in range(100):
            rew = torch.randn(n, device="cuda", generator=g, dtype=torch.float64).to(dtype)
            term = (torch.rand(n, device="cuda", generator=g) < 0.07).to(torch.uint8)
            trunc = (torch.rand(n, device="cuda", generator=g) < 0.03).to(torch.uint8)
            info = torch.randint(0, 9, (n, idim), device="cuda", generator=g, dtype=torch.int32)
            # the torch ops
            dones = (term | trunc).bool()
            a["cur_rew"].add_(rew.to(torch.float64)); a["cur_len"].add_(1); a["step"].add_(1)
            take = dones & (a["counts"] < tg)
            slot = a["counts"].clamp(max=E - 1)
            a["fin_rew"][ar, slot] = torch.where(take, a["cur_rew"], a["fin_rew"][ar, slot])
            a["fin_len"][ar, slot] = torch.where(take, a["cur_len"], a["fin_len"][ar, slot])
            a["fin_step"][ar, slot] = torch.where(take, a["step"].expand(n), a["fin_step"][ar, slot])
            a["fin_info"][ar, slot] = torch.where(take[:, None], info, a["fin_info"][ar, slot])
            a["counts"].add_(take.to(torch.int64))
            a["cur_rew"].masked_fill_(dones, 0.0); a["cur_len"].masked_fill_(dones, 0)
            # the launch
            rc = L.fw_eval_track(rew.data_ptr(), int(dtype == torch.float64), term.data_ptr(), trunc.data_ptr(), info.data_ptr(), idim,
                                 tg.data_ptr(), b["counts"].data_ptr(), b["cur_rew"].data_ptr(), b["cur_len"].data_ptr(), b["step"].data_ptr(),
                                 b["fin_rew"].data_ptr(), b["fin_len"].data_ptr(), b["fin_step"].data_ptr(), b["fin_info"].data_ptr(), n, E, None)
            assert rc == 0
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(a["counts"].sum()) > n and int(a["step"]) == 200
    assert L.fw_eval_track(None, 1, None, None, None, 0, None, None, None, None, None, None, None, None, None, n, E, None) == K.FW_EINVAL


def test_eval_harness_objlock_reports_duck_strike_rate():
    venv = P.FixedwingVecEnv(K.train_objlock_config(max_duration_seconds=3.0), 32, seed=1)
    env = R.VecNormalizeDevice(venv, training=False, norm_reward=False)
    pol = R.MlpPolicy(env.obs_dim).cuda()
    r = evaluate.evaluate_policy(pol, env, n_eval_episodes=32, deterministic=True)
    sc = r.scalars(has_duck=True)
    assert len(r.duck_strike) == 32 and 0.0 <= sc["eval/duck_strike_rate"] <= 1.0
    assert sc["eval/success_rate"] == pytest.approx(np.mean(r.is_success))


def _ppo(seed, graphs):
    venv = P.FixedwingVecEnv(_wp_cfg(), 256, seed=seed)
    env = R.VecNormalizeDevice(venv)
    return R.PPO(env, R.PPOConfig(n_steps=8, batch_size=256, n_epochs=2, seed=seed, use_graphs=graphs))


@pytest.mark.parametrize("graphs", [False, True])
def test_checkpoint_resume_continues_the_interrupted_episodes(tmp_path, graphs):
    a = _ppo(7, graphs)
    a.learn(3 * 8 * 256)
    path = checkpoint.save(str(tmp_path / "ck.pt"), a)
    state_at_save = a.env.venv.get_state().copy()
    a.learn(2 * 8 * 256, reset_num_timesteps=False)
    ref_params = [p.detach().clone() for p in a.policy.parameters()]
    ref_state = a.env.venv.get_state()

    b = _ppo(7, graphs)
    if graphs:
        b.learn(2 * 8 * 256)                                          # graphs captured BEFORE the load: it must update them in place
    sd = checkpoint.load(path, b, reset_num_timesteps=False, restore_env_state=True)
    np.testing.assert_array_equal(b.env.venv.get_state(), state_at_save)
    assert b.num_timesteps == sd["num_timesteps"] == 3 * 8 * 256
    b.learn(2 * 8 * 256, reset_num_timesteps=False)
    assert b.num_timesteps == 5 * 8 * 256
    tol = dict(rtol=0, atol=0) if not graphs else dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.env.venv.get_state(), ref_state, **(dict(rtol=0, atol=0) if not graphs else dict(rtol=1e-6, atol=1e-6)))
    for p, q in zip(ref_params, b.policy.parameters()):
        torch.testing.assert_close(q, p, **tol)


def test_train_script_flow_eval_best_model_checkpoints_and_final_save(tmp_path):
    """The callback wiring of train/train_Fixedwing_Waypoints_v3.py:271-347 end to end."""
    a = _ppo(3, True)
    eval_env = R.VecNormalizeDevice(P.FixedwingVecEnv(_wp_cfg(), 16, seed=3, global_env_offset=256), training=False, norm_reward=False)
    ev = evaluate.EvalCallback(eval_env, n_eval_episodes=16, eval_freq=16, log_path=str(tmp_path / "logs"),
                               best_model_save_path=str(tmp_path / "models"), num_targets_total=8)
    ck = checkpoint.CheckpointCallback(save_freq=16, save_path=str(tmp_path / "models"), name_prefix="waypoints_ppo")
    a.learn(4 * 8 * 256, callbacks=[ev, ck])
    assert ev.n_evals == 2 and len(ck.saved) == 2
    assert torch.equal(eval_env.obs_rms.mean, a.env.obs_rms.mean)                 # synced before evaluating
    final = checkpoint.save(str(tmp_path / "models" / "final_model.pt"), a)
    vn = checkpoint.save_vecnormalize(str(tmp_path / "models" / "vecnorm.pt"), a.env)
    assert checkpoint.infer_vecnorm_path(final, None) == vn
    b = _ppo(4, True)
    checkpoint.set_parameters(os.path.join(str(tmp_path / "models"), "best_model.pt"), b)
    checkpoint.load_vecnormalize(vn, b.env, training=True, norm_reward=True)
    b.learn(8 * 256)                                                              # fresh counter, pretrained weights
    assert b.num_timesteps == 8 * 256 and all(torch.isfinite(p).all() for p in b.policy.parameters())


def test_overlapped_evaluation_reports_what_the_synchronous_one_does(tmp_path):
    """EvalCallback's default on one GPU launches the evaluation on a side stream (copy of the weights, normaliser statistics,
    every replay enqueued up front) and lets the training go on; the figures it logs -- and the best_model it writes from the
    snapshot taken at launch -- are those of the callback that evaluates on the spot, for the same training run."""
    runs = []
    for overlap in (False, True):
        a = _ppo(5, True)
        eval_env = R.VecNormalizeDevice(P.FixedwingVecEnv(_wp_cfg(), 16, seed=5, global_env_offset=256), training=False, norm_reward=False)
        d = tmp_path / ("ov" if overlap else "sync")
        ev = evaluate.EvalCallback(eval_env, n_eval_episodes=20, eval_freq=16, log_path=str(d / "logs"),
                                   best_model_save_path=str(d / "models"), num_targets_total=8, overlap=overlap)
        a.learn(6 * 8 * 256, callbacks=[ev])
        assert ev.n_evals == 3 and ev._pending is None
        best = torch.load(os.path.join(str(d / "models"), "best_model.pt"), map_location="cpu", weights_only=True)
        runs.append((ev, best, [p.detach().cpu().clone() for p in a.policy.parameters()]))
    (e0, b0, p0), (e1, b1, p1) = runs
    assert e0.evaluations_timesteps == e1.evaluations_timesteps == [2 * 8 * 256, 4 * 8 * 256, 6 * 8 * 256]
    assert e0.evaluations_results == e1.evaluations_results and e0.evaluations_length == e1.evaluations_length
    assert e0.best_mean_reward == e1.best_mean_reward and b0["num_timesteps"] == b1["num_timesteps"]
    for k in b0["policy"]:
        assert torch.equal(b0["policy"][k], b1["policy"][k]), k
    for x, y in zip(p0, p1):                                                          # and the training itself did not notice
        assert torch.equal(x, y)
