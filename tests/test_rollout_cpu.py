"""CPU checks of the rollout-collector host logic against literal restatements of the SB3
algorithms the reference's train scripts rely on (train/train_Fixedwing_Waypoints_v3.py:260,
293-310), plus world_size-2 gloo tests of the update-time collectives."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from pyflyt_drone_amd import rollout as R


# ---------------------------------------------------------------- literal numpy restatements (checkers)
class NpRunningMeanStd:                      # SB3 common/running_mean_std.py
    def __init__(self, shape=(), epsilon=1e-4):
        self.mean, self.var, self.count = np.zeros(shape), np.ones(shape), epsilon

    def update(self, arr):
        bm, bv, bc = arr.mean(axis=0), arr.var(axis=0), arr.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


def np_gae(rewards, values, episode_starts, last_values, dones, gamma, lam):   # RolloutBuffer.compute_returns_and_advantage
    T = rewards.shape[0]
    adv = np.zeros_like(rewards)
    last = 0
    for step in reversed(range(T)):
        if step == T - 1:
            nnt, nv = 1.0 - dones, last_values
        else:
            nnt, nv = 1.0 - episode_starts[step + 1], values[step + 1]
        delta = rewards[step] + gamma * nv * nnt - values[step]
        last = delta + gamma * lam * nnt * last
        adv[step] = last
    return adv, adv + values


class FakeVenv:
    """CPU stand-in with the device-env surface VecNormalizeDevice consumes (tests only)."""

    def __init__(self, n=16, d=6, seed=0, horizon=7):
        self.device = torch.device("cpu"); self.num_envs, self.obs_dim = n, d
        self.torch_dtype = torch.float64
        self.g = torch.Generator().manual_seed(seed)
        self.t = torch.zeros(n, dtype=torch.long); self.horizon = horizon
        self.terminal_obs = torch.zeros((n, d), dtype=torch.float64)
        self.obs = torch.zeros((n, d), dtype=torch.float64)

    def _draw(self):
        return torch.randn((self.num_envs, self.obs_dim), generator=self.g, dtype=torch.float64) * 3.0 + 1.5

    def reset_tensor(self):
        self.t.zero_(); self.obs = self._draw(); return self.obs

    def step_tensor(self, actions):
        self.t += 1
        nxt = self._draw()
        rew = -(actions.to(torch.float64) ** 2).sum(-1) + 0.1 * nxt[:, 0]
        trunc = (self.t >= self.horizon + (torch.arange(self.num_envs) % 3))
        term = (nxt[:, 1] > 7.0) & ~trunc
        done = term | trunc
        self.terminal_obs = torch.where(done[:, None], nxt, self.terminal_obs)
        fresh = self._draw()
        self.obs = torch.where(done[:, None], fresh, nxt)
        self.t = torch.where(done, torch.zeros_like(self.t), self.t)
        return self.obs, rew, term.to(torch.uint8), trunc.to(torch.uint8)


# ---------------------------------------------------------------- tests
def test_running_mean_std_matches_sb3_formula():
    rng = np.random.default_rng(0)
    a, b = NpRunningMeanStd((5,)), R.RunningMeanStd((5,), "cpu")
    for n in (7, 1, 64, 300):
        x = rng.normal(2.0, 3.0, size=(n, 5))
        a.update(x); b.update(torch.as_tensor(x))
        np.testing.assert_allclose(b.mean.numpy(), a.mean, rtol=1e-12)
        np.testing.assert_allclose(b.var.numpy(), a.var, rtol=1e-11)
        assert float(b.count) == pytest.approx(a.count)
    s = R.RunningMeanStd((), "cpu"); s.update(torch.as_tensor(rng.normal(size=100)))
    assert s.mean.shape == ()


def test_gae_reference_matches_sb3_loop():
    rng = np.random.default_rng(1)
    T, N = 37, 11
    r, v = rng.normal(size=(T, N)).astype(np.float32), rng.normal(size=(T, N)).astype(np.float32)
    es = (rng.uniform(size=(T, N)) < 0.1).astype(np.float32)
    lv, d = rng.normal(size=N).astype(np.float32), (rng.uniform(size=N) < 0.3).astype(np.float32)
    a0, r0 = np_gae(r, v, es, lv, d, 0.99, 0.95)
    a1, r1 = R.gae_reference(*(torch.as_tensor(x) for x in (r, v, es, lv, d)), 0.99, 0.95)
    np.testing.assert_allclose(a1.numpy(), a0, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r1.numpy(), r0, rtol=1e-5, atol=1e-6)


def test_policy_distribution_matches_torch_normal():
    torch.manual_seed(0)
    pol = R.MlpPolicy(9)
    assert sum(p.numel() for p in pol.parameters()) == 2 * (9 * 64 + 64 + 64 * 64 + 64) + 64 * 4 + 4 + 64 + 1 + 4
    obs = torch.randn(13, 9)
    with torch.no_grad():
        pol.log_std.copy_(torch.tensor([0.1, -0.3, 0.0, 0.5]))
        act, val, logp = pol(obs)
        mean = pol.action_net(pol.pi_net(obs))
        dist = torch.distributions.Normal(mean, pol.log_std.exp())
        torch.testing.assert_close(logp, dist.log_prob(act).sum(-1))
        v2, lp2, ent = pol.evaluate_actions(obs, act)
        torch.testing.assert_close(lp2, logp); torch.testing.assert_close(v2, val)
        torch.testing.assert_close(ent, dist.entropy().sum(-1))
        a_det, _, _ = pol(obs, deterministic=True)
        torch.testing.assert_close(a_det, mean)
    # SB3 init: action head gain 0.01 => near-zero mean actions, log_std 0
    fresh = R.MlpPolicy(9)
    assert fresh.action_net.weight.abs().max() < 0.05 and float(fresh.log_std.abs().max()) == 0.0


def test_vecnormalize_semantics_match_sb3():
    venv = FakeVenv(n=8, d=4, seed=3)
    vn = R.VecNormalizeDevice(venv, gamma=0.99, use_fused_kernel=False)
    obs_rms, ret_rms, returns = NpRunningMeanStd((4,)), NpRunningMeanStd(()), np.zeros(8)
    twin = FakeVenv(n=8, d=4, seed=3)
    o = twin.reset_tensor().numpy(); obs_rms.update(o)
    got = vn.reset()
    np.testing.assert_allclose(got.numpy(), np.clip((o - obs_rms.mean) / np.sqrt(obs_rms.var + 1e-8), -10, 10), rtol=1e-5, atol=1e-6)
    g = torch.Generator().manual_seed(9)
    for _ in range(25):
        a = torch.randn((8, 4), generator=g)
        o, r, te, tr = twin.step_tensor(a)
        o, r = o.numpy(), r.numpy(); done = (te | tr).numpy().astype(bool)
        obs_rms.update(o)
        returns = returns * 0.99 + r; ret_rms.update(returns)
        want_r = np.clip(r / np.sqrt(ret_rms.var + 1e-8), -10, 10)
        returns[done] = 0
        on, rn, dn, to, tobs = vn.step(a)
        np.testing.assert_allclose(on.numpy(), np.clip((o - obs_rms.mean) / np.sqrt(obs_rms.var + 1e-8), -10, 10), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(rn.numpy(), want_r, rtol=1e-5, atol=1e-6)
        assert np.array_equal(dn.numpy(), done) and np.array_equal(to.numpy(), (tr.numpy() == 1) & (te.numpy() == 0))
        want_t = np.clip((twin.terminal_obs.numpy() - obs_rms.mean) / np.sqrt(obs_rms.var + 1e-8), -10, 10)
        np.testing.assert_allclose(tobs.numpy()[done], want_t[done], rtol=1e-5, atol=1e-5)
    # eval mode: statistics frozen, reward untouched (eval env of the train script: training=False, norm_reward=False)
    ev = R.VecNormalizeDevice(FakeVenv(n=8, d=4, seed=5), training=False, norm_reward=False, use_fused_kernel=False)
    ev.load_state_dict(vn.state_dict()); ev.training, ev.norm_reward = False, False
    before = ev.obs_rms.mean.clone(); ev.reset(); _, r2, *_ = ev.step(torch.zeros(8, 4))
    assert torch.equal(before, ev.obs_rms.mean) and r2.abs().max() < 3.0


def test_ppo_runs_on_fake_env_and_improves_objective():
    venv = FakeVenv(n=32, d=5, seed=1)
    env = R.VecNormalizeDevice(venv, use_fused_kernel=False)
    cfg = R.PPOConfig(n_steps=8, batch_size=64, n_epochs=4, seed=7)
    ppo = R.PPO(env, cfg, policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
    w0 = [p.detach().clone() for p in ppo.policy.parameters()]
    ppo.learn(total_timesteps=3 * 8 * 32)
    assert ppo.num_timesteps == 3 * 8 * 32
    assert all(math.isfinite(v) for v in ppo.logs.values())
    assert any(not torch.equal(a, b) for a, b in zip(w0, ppo.policy.parameters()))
    # truncation bootstrap: reward of a timed-out step includes gamma * V(terminal_obs)
    sd = ppo.state_dict()
    ppo2 = R.PPO(R.VecNormalizeDevice(FakeVenv(n=32, d=5, seed=1), use_fused_kernel=False), cfg, policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
    ppo2.load_state_dict(sd)
    assert ppo2.num_timesteps == 0                      # the reference restarts the counter on resume (:313-327)
    for a, b in zip(ppo.policy.parameters(), ppo2.policy.parameters()):
        assert torch.equal(a, b)
    assert torch.equal(ppo.env.obs_rms.mean, ppo2.env.obs_rms.mean)


# ---------------------------------------------------------------- gloo, world_size 2
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(123)
        full = torch.randn((40, 6), generator=g, dtype=torch.float64) * 2 + 1
        mine = full[rank * 20:(rank + 1) * 20]
        # (1) normaliser moments: merged statistics == single-process statistics of the union
        rms = R.RunningMeanStd((6,), "cpu"); rms.update(mine)
        ref = NpRunningMeanStd((6,)); ref.update(full.numpy())
        ok1 = np.allclose(rms.mean.numpy(), ref.mean, rtol=1e-12) and np.allclose(rms.var.numpy(), ref.var, rtol=1e-11)
        # (2) advantage all-gather == statistics of the union == the 2-scalar all-reduce formula
        adv_full = torch.randn(64, generator=g); adv = adv_full[rank * 32:(rank + 1) * 32]
        m, s, allv = R.global_advantage_stats(adv)
        buf = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()))]); td.all_reduce(buf)
        m2 = buf[0] / buf[2]; s2 = torch.sqrt((buf[1] - buf[2] * m2 * m2) / (buf[2] - 1))
        ok2 = (torch.allclose(m, adv_full.mean()) and torch.allclose(s, adv_full.std()) and torch.equal(allv, adv_full)
               and torch.allclose(m, m2, atol=1e-6) and torch.allclose(s, s2, atol=1e-5))
        # (3) one flattened gradient bucket: every rank ends with the mean gradient
        lin = torch.nn.Linear(3, 2)
        for p in lin.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        R.allreduce_grads_(list(lin.parameters()))
        ok3 = all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in lin.parameters())
        # (4) a sharded PPO job keeps its replicas identical and counts global timesteps
        venv = FakeVenv(n=8, d=5, seed=100 + rank)
        ppo = R.PPO(R.VecNormalizeDevice(venv, use_fused_kernel=False), R.PPOConfig(n_steps=4, batch_size=16, n_epochs=2, seed=3),
                    policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
        ppo.learn(total_timesteps=2 * 4 * 8 * world)
        flat = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()])
        others = [torch.empty_like(flat) for _ in range(world)]; td.all_gather(others, flat)
        ok4 = all(torch.allclose(o, flat, atol=1e-7) for o in others) and ppo.num_timesteps == 2 * 4 * 8 * world
        stats = torch.cat([ppo.env.obs_rms.mean, ppo.env.obs_rms.var]); so = [torch.empty_like(stats) for _ in range(world)]
        td.all_gather(so, stats)
        ok5 = all(torch.allclose(o, stats) for o in so)
        q.put((rank, ok1, ok2, ok3, ok4, ok5))
    finally:
        td.destroy_process_group()


def test_update_time_collectives_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    for rank, *oks in sorted(res):
        assert all(oks), (rank, oks)


# ---------------------------------------------------------------- replicated update of a sharded job (gloo, world_size 2)
class RecordingVenv(FakeVenv):
    """FakeVenv that keeps every raw observation batch it handed out (reset and steps), for the statistics check."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw); self.seen = []

    def reset_tensor(self):
        o = super().reset_tensor(); self.seen.append(o.clone()); return o

    def step_tensor(self, actions):
        out = super().step_tensor(actions); self.seen.append(out[0].clone()); return out


_REPL_CFG = dict(n_steps=4, batch_size=16, n_epochs=3, seed=3)


def _worker_replicated(rank, world, port, q, stats_sync):
    import torch.distributed as td
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        venv = RecordingVenv(n=8, d=5, seed=100 + rank)
        env = R.VecNormalizeDevice(venv, use_fused_kernel=False, stats_sync=stats_sync)
        ppo = R.PPO(env, R.PPOConfig(**_REPL_CFG), policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
        assert ppo._replicated and ppo.perm_gen is not ppo.gen
        w0 = [p.detach().clone().numpy() for p in ppo.policy.parameters()]
        ppo.collect_rollouts()
        shard = {k: getattr(ppo, k).clone().numpy() for k in ("buf_obs", "buf_act", "buf_logp", "adv", "ret")}
        stats = torch.cat([env.obs_rms.mean, env.obs_rms.var, env.obs_rms.count, env.ret_rms.mean.reshape(1),
                           env.ret_rms.var.reshape(1), env.ret_rms.count]).numpy()
        ppo.train()
        w1 = [p.detach().clone().numpy() for p in ppo.policy.parameters()]
        chk = ppo.replica_checksum()
        # two more iterations: the replicas must stay bit-identical with no per-minibatch collective at all
        ppo.learn(2 * 4 * 8 * world, reset_num_timesteps=False)
        chk2 = ppo.replica_checksum()
        seen = torch.cat(venv.seen[:5]).numpy()                  # reset + the 4 steps of the first rollout
        q.put((rank, w0, shard, stats, w1, chk, chk2, ppo.allgather_bytes, seen, ppo.num_timesteps))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("stats_sync", ["step", "rollout"])
def test_replicated_update_world_size_2_equals_single_process_on_concatenated_buffers(stats_sync):
    """SURVEY section 8(e) / north_star: envs shard across ranks, the rollout shard is all-gathered at update time and every
    rank runs the same minibatch sequence.  Proven here on the CPU (gloo, torch restatement of the learner): (a) both ranks
    end with bit-identical weights, (b) those are bit-identical to ONE process running PPO.train() on the concatenated
    buffers, (c) the normaliser statistics of the ranks agree bit for bit and equal the statistics of the union of all
    observations, for both exchange cadences (every vec-step / once per rollout)."""
    import torch.multiprocessing as mp
    torch.set_num_threads(1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_replicated, args=(r, 2, port, q, stats_sync)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    (_, w0a, sa, sta, w1a, chka, chk2a, nbytes, seen_a, nt_a), (_, w0b, sb, stb, w1b, chkb, chk2b, _, seen_b, nt_b) = res
    assert all(np.array_equal(x, y) for x, y in zip(w0a, w0b)), "rank 0's initial weights are broadcast"
    assert all(np.array_equal(x, y) for x, y in zip(w1a, w1b)), "replicas diverged"
    assert chka == chkb == 0.0 and chk2a == chk2b == 0.0
    assert nbytes == 4 * 8 * (5 + 7) * 4 and nt_a == nt_b == 3 * 4 * 8 * 2
    assert np.array_equal(sta, stb), "normaliser statistics differ between ranks"
    ref = NpRunningMeanStd((5,))
    if stats_sync == "step":
        for k in range(5):                                            # one merge per vec-step over the union, as one big VecNormalize
            ref.update(np.concatenate([seen_a[8 * k:8 * k + 8], seen_b[8 * k:8 * k + 8]]))
    else:
        ref.update(np.concatenate([seen_a, seen_b]))                  # one merge of everything since the last exchange
    np.testing.assert_allclose(sta[:5], ref.mean, rtol=1e-12); np.testing.assert_allclose(sta[5:10], ref.var, rtol=1e-11)
    assert sta[10] == pytest.approx(1e-4 + 5 * 16)
    # single process on the concatenated buffers (rank-major env order), same initial weights, same permutation stream
    one = R.PPO(R.VecNormalizeDevice(FakeVenv(n=16, d=5, seed=0), use_fused_kernel=False), R.PPOConfig(**_REPL_CFG),
                policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
    with torch.no_grad():
        for p, w in zip(one.policy.parameters(), w0a):
            p.copy_(torch.from_numpy(w))
    for k in ("buf_obs", "buf_act", "buf_logp"):
        getattr(one, k).copy_(torch.from_numpy(np.concatenate([sa[k], sb[k]], axis=1)))
    one.adv = torch.from_numpy(np.concatenate([sa["adv"], sb["adv"]], axis=1)); one.ret = torch.from_numpy(np.concatenate([sa["ret"], sb["ret"]], axis=1))
    one.perm_gen = torch.Generator().manual_seed(_REPL_CFG["seed"] * 1_000_003 + 12345)
    one.train()
    for p, w in zip(one.policy.parameters(), w1a):
        assert np.array_equal(p.detach().numpy(), w), "sharded replicas != one process on the concatenated buffers"


# ---------------------------------------------------------------- configs[0]: CPU plumbing
class OracleVenv:
    """Test-only adapter: the CPU oracle behind the device-env surface the learner consumes.  (The product has no CPU
    env; this is BASELINE.json configs[0] -- "single env on CPU via the train script, plumbing" -- with the oracle in the
    place of PyBullet, which cannot be installed here.)"""

    def __init__(self, oracle, cfg, n, seed):
        from pyflyt_drone_amd import config as K
        self.env = oracle.OracleEnv(cfg, n, seed=seed)
        self.device, self.num_envs, self.obs_dim = torch.device("cpu"), n, K.obs_dim(cfg)
        self.torch_dtype = torch.float64
        self.terminal_obs = torch.zeros((n, self.obs_dim), dtype=torch.float64)
        self.rewards = torch.zeros(n, dtype=torch.float64)

    def reset_tensor(self):
        return torch.from_numpy(np.asarray(self.env.reset(), dtype=np.float64).copy())

    def step_tensor(self, actions):
        o, r, te, tr, to, info = self.env.step(actions.numpy())
        self.terminal_obs = torch.from_numpy(np.asarray(to, dtype=np.float64).copy())
        self.rewards = torch.from_numpy(np.asarray(r, dtype=np.float64).copy())
        return (torch.from_numpy(np.asarray(o, dtype=np.float64).copy()), self.rewards,
                torch.from_numpy(np.asarray(te, dtype=np.uint8).copy()), torch.from_numpy(np.asarray(tr, dtype=np.uint8).copy()))


def test_config0_plumbing_oracle_env_driven_by_the_ppo_for_two_updates(oracle, K):
    """train/train_Fixedwing_Waypoints_v3.py hyper-parameters (:27-55: batch 128, 20 -> 2 epochs here, lr 3e-4, ent 0.001,
    num_targets 8, sparse, euler, 30 Hz) on a single CPU env: two updates, finite, counters right, fps reported."""
    import time
    cfg = K.train_waypoints_v3_config()
    venv = OracleVenv(oracle, cfg, 1, seed=42)
    env = R.VecNormalizeDevice(venv, use_fused_kernel=False)
    ppo = R.PPO(env, R.PPOConfig(n_steps=256, batch_size=128, n_epochs=2, learning_rate=3e-4, ent_coef=0.001, seed=42,
                                 use_graphs=False, fused_update=False, fused_collect=False), gae_fn=R.gae_reference)
    t0 = time.perf_counter()
    ppo.learn(2 * 256)
    fps = ppo.num_timesteps / (time.perf_counter() - t0)
    assert ppo.num_timesteps == 512 and fps > 0
    assert all(torch.isfinite(p).all() for p in ppo.policy.parameters())
    assert all(math.isfinite(v) for v in ppo.logs.values())
    assert float(env.obs_rms.count) == pytest.approx(1e-4 + 513) and torch.isfinite(env.obs_rms.var).all()
    print(f"configs[0] plumbing: {fps:.0f} env-steps/s (1 CPU env, oracle physics, torch PPO)")


# ---------------------------------------------------------------- CNN detector head (BASELINE configs[4]) on the torch path
class ImageVenv(FakeVenv):
    """FakeVenv with the camera surface (`render_tensor`): the image is a function of the current observation, and the reward
    pays for steering towards a quantity that is ONLY visible in the image (its bright column)."""

    def _col(self):
        return (self.obs[:, 2].abs() * 3).long() % 8

    def render_tensor(self, res, out=None):
        n = self.num_envs
        img = torch.zeros((n, 2, res, res), dtype=torch.float32) if out is None else out
        img.zero_()
        img[torch.arange(n), 0, :, self._col()] = 1.0
        img[:, 1] = 0.5
        return img

    def step_tensor(self, actions):
        want = (self._col().to(torch.float64) - 3.5) / 3.5              # the image's bright column, as an action target
        obs, rew, term, trunc = super().step_tensor(actions)
        rew = -(actions[:, 0].to(torch.float64) - want) ** 2
        return obs, rew, term, trunc


def test_cnn_detector_policy_shapes_and_wiring():
    torch.manual_seed(0)
    pol = R.CnnDetectorPolicy(6, image_res=16, cnn_features=8)
    obs, img = torch.randn(5, 6), torch.rand(5, 2, 16, 16)
    a, v, lp = pol(obs, img=img)
    assert a.shape == (5, 4) and v.shape == (5,) and lp.shape == (5,)
    v2, lp2, ent = pol.evaluate_actions(obs, a, img=img)
    assert torch.allclose(v2, v) and torch.allclose(lp2, lp, atol=1e-6) and ent.shape == (5,)
    with pytest.raises(ValueError):
        pol(obs)                                                          # the image is not optional
    assert not R.FusedPpoUpdate.fits(pol, 6, torch.device("cpu")) and R.policy_inputs(R.MlpPolicy(6), None) == {}
    # the extractor feeds BOTH networks: a value-only loss reaches the conv weights
    pol.zero_grad(); pol.predict_values(obs, img=img).sum().backward()
    assert pol.cnn[0].weight.grad.abs().sum() > 0


def test_ppo_with_cnn_front_end_collects_images_and_trains_through_the_extractor():
    venv = ImageVenv(n=32, d=6, seed=4, horizon=50)
    env = R.VecNormalizeDevice(venv, use_fused_kernel=False, norm_reward=False)
    ppo = R.PPO(env, R.PPOConfig(n_steps=8, batch_size=64, n_epochs=4, learning_rate=3e-3, detector="cnn", image_res=8, cnn_features=8, seed=1),
                gae_fn=R.gae_reference)
    assert isinstance(ppo.policy, R.CnnDetectorPolicy) and not ppo._collect_fused and not ppo._graphs and ppo._fused is None
    w0 = ppo.policy.cnn[0].weight.detach().clone()
    ppo.collect_rollouts()
    assert ppo.buf_img.shape == (8, 32, 2, 8, 8)
    assert float(ppo.buf_img[:, :, 0].sum(dim=(-1, -2)).min()) == 8.0     # every stored image holds its bright column
    first = float(ppo.buf_rew.mean())
    ppo.train()
    assert not torch.equal(w0, ppo.policy.cnn[0].weight) and all(np.isfinite(v) for v in ppo.logs.values())
    ppo.learn(60 * 8 * 32, reset_num_timesteps=False)
    ppo.collect_rollouts()
    assert float(ppo.buf_rew.mean()) > first + 0.1, (first, float(ppo.buf_rew.mean()))    # the cue is only in the image: it was used


def _worker_cnn(rank, world, port, q):
    import torch.distributed as td
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = R.VecNormalizeDevice(ImageVenv(n=8, d=6, seed=50 + rank, horizon=9), use_fused_kernel=False, stats_sync="step")
        ppo = R.PPO(env, R.PPOConfig(n_steps=4, batch_size=16, n_epochs=2, detector="cnn", image_res=8, cnn_features=8, seed=2), gae_fn=R.gae_reference)
        facts = (ppo._replicated, ppo._img)
        ppo.learn(3 * 4 * 8 * world)
        flat = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()])
        q.put((rank, facts, ppo.replica_checksum(), flat.numpy(), ppo.num_timesteps))
    finally:
        td.destroy_process_group()


def test_cnn_front_end_world_size_2_all_reduces_gradients_and_keeps_the_replicas_identical():
    """configs[4] shards its envs over GPUs with a CNN policy: that job is data-parallel (local minibatches, one flattened
    gradient all-reduce each) -- the rollout all-gather of the MLP jobs would move every rank's images to every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_cnn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    (_, fa, ca, wa, na), (_, fb, cb, wb, nb) = res
    assert fa == fb == (False, True)
    assert ca == cb == 0.0 and np.array_equal(wa, wb) and na == nb == 3 * 4 * 8 * 2


# ---------------------------------------------------------------- samples per update are held when the envs multiply
def test_n_steps_for_holds_the_reference_samples_per_update():
    assert R.n_steps_for(32 * 2048, 4096, 1) == 16 and R.n_steps_for(32 * 2048, 4096, 8) == 2        # configs[3]: 8 x 4096 envs
    assert R.n_steps_for(32 * 1024, 2048, 8) == 2 and R.n_steps_for(16 * 2048, 4096, 1) == 8        # configs[4] / configs[2]
    assert R.n_steps_for(100, 4096, 8) == 1                                                         # never below one step
    assert R.init_distributed_from_env() == (1, 0, 0) or os.environ.get("WORLD_SIZE", "1") != "1"


def _worker_samples(rank, world, port, q):
    import torch.distributed as td
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      FW_DIST_BACKEND="gloo")
    w, r, _ = R.init_distributed_from_env()                      # what the examples call first
    try:
        S, n = 64, 8                                             # samples per update of the (toy) reference config, envs per rank
        T = R.n_steps_for(S, n, w)
        env = R.VecNormalizeDevice(FakeVenv(n=n, d=5, seed=7 + r), use_fused_kernel=False, stats_sync="rollout")
        ppo = R.PPO(env, R.PPOConfig(n_steps=T, batch_size=16, n_epochs=1, seed=3), policy=R.MlpPolicy(5), gae_fn=R.gae_reference)
        ppo.collect_rollouts()
        B = ppo._update_buffers()[0].shape[0]
        q.put((r, w, T, B, ppo.num_timesteps))
    finally:
        td.destroy_process_group()


def test_sharded_job_holds_the_samples_per_update_world_size_2():
    """in a sharded job the update is replicated (every rank walks ALL gathered samples), so the examples
    divide n_steps by the world size: the update sees the reference's sample count whatever the number of GPUs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_samples, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    for r, w, T, B, nt in res:
        assert w == 2 and T == 4 and B == 64 and nt == 64            # 8 envs x 2 ranks x 4 steps = the 64 samples of one process with 8 steps
