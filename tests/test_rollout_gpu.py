"""GPU checks of the collector kernels (through the C ABI) and an end-to-end PPO run on the
fused env -- configs[0]/[1] plumbing: the hyper-parameters of
train/train_Fixedwing_Waypoints_v3.py:27-55 with n_steps scaled to the env count."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
from pyflyt_drone_amd import rollout as R

pytestmark = pytest.mark.gpu


def test_fw_gae_matches_reference():
    g = torch.Generator().manual_seed(0)
    for T, N in ((16, 4096), (37, 1000), (2048, 33)):
        r, v = torch.randn((T, N), generator=g), torch.randn((T, N), generator=g)
        es = (torch.rand((T, N), generator=g) < 0.05).float()
        lv, d = torch.randn(N, generator=g), (torch.rand(N, generator=g) < 0.2).float()
        a0, r0 = R.gae_reference(r, v, es, lv, d, 0.99, 0.95)
        a1, r1 = R.gae_device(*(x.cuda() for x in (r, v, es, lv, d)), 0.99, 0.95)
        torch.testing.assert_close(a1.cpu(), a0, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(r1.cpu(), r0, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fw_normalize_obs_matches_torch_path(dtype):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1)

    class V:                     # minimal venv surface
        device, num_envs, obs_dim = dev, 4096, 28
    fused = R.VecNormalizeDevice(V(), use_fused_kernel=True)
    plain = R.VecNormalizeDevice(V(), use_fused_kernel=False)
    # the accumulators a sharded job all-reduces once per rollout (fw_normalize_obs's batch_acc): sums of every UPDATING batch
    fused._obs_acc = torch.zeros(2 * 28 + 1, dtype=torch.float64, device=dev)
    want = torch.zeros_like(fused._obs_acc)
    for i in range(6):
        obs = (torch.randn((4096, 28), generator=g, dtype=torch.float64) * (1 + i) + 3 * i).to(dtype).to(dev)
        if i != 4:
            x = obs.double()
            want += torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([4096.0], dtype=torch.float64, device=dev)])
        a = fused._process_obs(obs, update=(i != 4)).clone()
        torch.testing.assert_close(fused._obs_acc, want, rtol=1e-12, atol=1e-9)
        b = plain._process_obs(obs, update=(i != 4)).clone()
        torch.testing.assert_close(fused.obs_rms.mean, plain.obs_rms.mean, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(fused.obs_rms.var, plain.obs_rms.var, rtol=1e-9, atol=1e-12)
        torch.testing.assert_close(fused.obs_rms.count, plain.obs_rms.count)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
        assert a.abs().max() <= 10.0


def test_ppo_end_to_end_on_device_env():
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 1024, seed=42)
    vn = R.VecNormalizeDevice(env, norm_obs=True, norm_reward=True, clip_obs=10.0)       # :260
    cfg = R.PPOConfig(n_steps=16, batch_size=128, n_epochs=2, learning_rate=3e-4, gamma=0.99, gae_lambda=0.95,
                      clip_range=0.2, ent_coef=0.001, vf_coef=0.5, max_grad_norm=0.5, seed=42)     # :293-310
    ppo = R.PPO(vn, cfg)
    ppo.collect_rollouts()
    assert bool((ppo.buf_start[0] == 1).all()) and float(ppo.buf_start[1:].sum()) == 0.0    # every env starts an episode at t=0
    ppo.train()
    ppo.learn(total_timesteps=4 * 16 * 1024)
    assert ppo.num_timesteps >= 4 * 16 * 1024
    assert all(math.isfinite(v) for v in ppo.logs.values()), ppo.logs
    assert torch.isfinite(ppo.adv).all() and torch.isfinite(ppo.buf_obs).all()
    assert ppo.buf_obs.abs().max() <= 10.0 and float(vn.obs_rms.count) > 16 * 1024
    # checkpoint round trip (model + vecnormalize), counter reset like the reference's resume
    sd = ppo.state_dict()
    ppo2 = R.PPO(R.VecNormalizeDevice(P.FixedwingVecEnv(K.train_waypoints_v3_config(), 1024, seed=42)), cfg)
    ppo2.load_state_dict(sd)
    torch.testing.assert_close(ppo2.env.obs_rms.mean, vn.obs_rms.mean)
    assert ppo2.num_timesteps == 0


def test_collect_rollouts_is_graph_capturable():
    """fw_step + fw_normalize_obs + the policy forward run on one stream with no host sync."""
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 4096, seed=1)
    vn = R.VecNormalizeDevice(env)
    pol = R.MlpPolicy(env.obs_dim).cuda()
    obs = vn.reset().clone()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)

    def one_step():
        with torch.no_grad():
            a, v, lp = pol(obs, deterministic=True)
            o, r, d, to, tob = vn.step(a.clamp(-1, 1).double())
            obs.copy_(o)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            one_step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one_step()
    c0 = float(vn.obs_rms.count)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    assert float(vn.obs_rms.count) == pytest.approx(c0 + 5 * 4096) and torch.isfinite(obs).all()
    # the reward statistics accumulate across replays too (they are updated in place, not rebound)
    assert float(vn.ret_rms.count) == pytest.approx(1e-4 + (3 + 5) * 4096)     # capture itself executes nothing


def test_graph_replayed_rollouts_accumulate_the_same_statistics_as_eager_ones():
    """4 rollouts through PPO.collect_rollouts with and without hipGraph replay: identical sample counts in both
    normalisers, and the discounted-return tracker carries over from one replay to the next."""
    out = {}
    for graphs in (False, True):
        env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 512, seed=2)
        ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=8, batch_size=512, n_epochs=1, seed=2, use_graphs=graphs))
        for _ in range(4):
            ppo.collect_rollouts()
        torch.cuda.synchronize()
        out[graphs] = (float(ppo.env.obs_rms.count), float(ppo.env.ret_rms.count), ppo.env.returns.abs().max().item(),
                       float(ppo.env.ret_rms.var))
    assert out[True][0] == pytest.approx(1e-4 + (4 * 8 + 1) * 512) == pytest.approx(out[False][0])
    assert out[True][1] == pytest.approx(1e-4 + 4 * 8 * 512) == pytest.approx(out[False][1])
    assert out[True][2] > 1.0 and out[False][2] > 1.0              # ~32 steps of -0.1 discounted: the tracker was not restarted
    assert out[True][3] == pytest.approx(out[False][3], rel=0.2)


class _BufEnv:
    """Just enough env for PPO.__init__ / train(): the update is tested on hand-filled rollout buffers."""
    def __init__(self, n, d):
        self.device, self.num_envs, self.obs_dim = torch.device("cuda"), n, d


def _filled_ppo(fused, d, bs, n_epochs, T=4, n=256, seed=5, scope="minibatch", steps_before=0):
    ppo = R.PPO(_BufEnv(n, d), R.PPOConfig(n_steps=T, batch_size=bs, n_epochs=n_epochs, seed=seed, use_graphs=False,
                                           fused_update=fused, adv_norm_scope=scope, ent_coef=0.01))
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    ppo.buf_obs.copy_(torch.randn(ppo.buf_obs.shape, device="cuda", generator=g).clamp(-10, 10))
    with torch.no_grad():
        a, v, lp = ppo.policy(ppo.buf_obs.reshape(-1, d), generator=g)
        ppo.buf_act.copy_((a + 0.3 * torch.randn(a.shape, device="cuda", generator=g)).reshape(ppo.buf_act.shape))   # off-policy enough to hit the clip
        _, lp2, _ = ppo.policy.evaluate_actions(ppo.buf_obs.reshape(-1, d), ppo.buf_act.reshape(-1, 4))
        ppo.buf_logp.copy_((lp2 + 0.2 * torch.randn(lp2.shape, device="cuda", generator=g)).reshape(T, n))
    ppo.adv = torch.randn((T, n), device="cuda", generator=g) * 2.0 + 0.5
    ppo.ret = torch.randn((T, n), device="cuda", generator=g) * 3.0
    return ppo


@pytest.mark.parametrize("d,bs,scope", [(28, 128, "minibatch"), (56, 64, "minibatch"), (28, 64, "global"), (27, 256, "minibatch"),
                                        (5, 64, "minibatch"),        # a row narrower than two float4s, not 16-byte aligned
                                        (28, 192, "minibatch"),      # three chunks per minibatch: the chunk halves run 2 and 1 of them
                                        (64, 128, "global"),         # the widest input: four dW1 tiles, no split
                                        (28, 96, "minibatch"),       # three 32-sample chunks over two blocks
                                        (28, 32, "minibatch"),       # two 16-sample passes on two blocks
                                        (28, 48, "minibatch"),       # three 16-sample passes on two blocks
                                        (28, 16, "minibatch"),       # one pass, one block per network: no swap
                                        (40, 512, "minibatch")])     # 64-sample chunks over four blocks, 2 each; K = 64 rows of W1 in the 32-sample form
@pytest.mark.parametrize("split", [None, "64x2", "32x2", "64x1", "16x4", "16x1", "32x4", "16x8", "32x8", "64x8", "all-to-all"])
def test_fused_ppo_update_matches_the_torch_path(d, bs, scope, split, monkeypatch):
    """fw_ppo_update (one kernel for the whole minibatch sequence) against the plain torch PPO.train() on the same
    buffers, permutations, initial weights and Adam state: parameters, Adam moments and step count agree to fp32
    rounding after 2 epochs (16-32 sequential minibatch steps), and again after a second call (warm Adam state)."""
    if split == "all-to-all":                       # four blocks per network with the gradient swap of the two-block cut (default: reduce-scatter + weight all-gather)
        if bs < 128:
            pytest.skip("fewer than four blocks per network")
        monkeypatch.setenv("FWSIM_PPO_RS", "0")
    elif split is not None:                         # dev knob: every cut of a minibatch (samples per pass x blocks per network) the kernel has
        ch, ns = (int(x) for x in split.split("x"))
        if bs % ch or bs // ch < ns:
            pytest.skip("this cut does not divide the minibatch")
        monkeypatch.setenv("FWSIM_PPO_SPLIT", split)
    T_ = 3 if bs in (192, 96, 48) else 4            # (3 x 256 samples divide into 192- / 96- / 48-sample minibatches)
    a, b = _filled_ppo(True, d, bs, 2, T=T_, scope=scope), _filled_ppo(False, d, bs, 2, T=T_, scope=scope)
    for p, q in zip(a.policy.parameters(), b.policy.parameters()):
        assert torch.equal(p, q)
    for rnd in range(2):
        a.train(); b.train()
        assert a._fused is not None and b._fused is None
        for (na, p), (nb_, q) in zip(a.policy.named_parameters(), b.policy.named_parameters()):
            torch.testing.assert_close(p, q, rtol=2e-3, atol=2e-5, msg=lambda m: f"{na} round {rnd}: {m}")
            sa, sb = a.optimizer.state[p], b.optimizer.state[q]
            assert float(sa["step"]) == float(sb["step"]) == (rnd + 1) * 2 * (T_ * 256 // bs)
            torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=5e-3, atol=1e-6)
            torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=5e-3, atol=1e-9)
        for k in ("policy_loss", "value_loss"):
            assert a.logs[k] == pytest.approx(b.logs[k], rel=2e-3, abs=1e-5)
    # the update moved the policy and did not blow up
    assert all(torch.isfinite(p).all() for p in a.policy.parameters())


def _edit_behind_ppos_back(ppo, edit):
    """The same deterministic edit for a fused and a torch-path twin -- each the way a user would do it, none of them through PPO."""
    pol, opt = ppo.policy, ppo.optimizer
    with torch.no_grad():
        if edit == "param_add":
            pol.action_net.bias.add_(0.05)
            pol.pi_net[0].weight.mul_(0.9)
        elif edit == "policy_load_state_dict":
            sd = {k: (v * 0.8 + 0.01) for k, v in pol.state_dict().items()}
            pol.load_state_dict(sd)
        elif edit == "exp_avg":
            opt.state[pol.vf_net[2].weight]["exp_avg"].mul_(-3.0)
            opt.state[pol.log_std]["exp_avg_sq"].add_(1e-3)
        elif edit == "optimizer_load_state_dict":
            sd = opt.state_dict()
            for st in sd["state"].values():
                st["exp_avg"] = st["exp_avg"] * 0.5
            opt.load_state_dict(sd)                                   # (replaces the state tensors: new addresses, version 0)
        elif edit == "data_write_plus_touch":
            pol.value_net.bias.data.add_(0.2)                         # invisible to the version counter by design of torch ...
            ppo.touch()                                               # ... so the caller says so
        else:
            raise AssertionError(edit)


@pytest.mark.parametrize("edit", ["param_add", "policy_load_state_dict", "exp_avg", "optimizer_load_state_dict", "data_write_plus_touch"])
def test_edits_of_module_or_optimiser_between_two_updates_are_seen_by_the_fused_update(edit):
    """Round 4 let fw_ppo_update reuse the flat parameter / moment images of its previous call unless PPO itself had cleared a
    flag: anything touching ``ppo.policy`` / ``ppo.optimizer`` behind PPO's back was silently ignored.  The images are now guarded
    by the tensors' own version counters and addresses: after any such edit the fused update equals the torch path started from
    the EDITED state (it would differ by the size of the edit otherwise)."""
    a, b = _filled_ppo(True, 28, 128, 2), _filled_ppo(False, 28, 128, 2)
    a.train(); b.train()
    assert a._fused is not None and a._fused.synced and b._fused is None
    sig = a._fused.state_signature()
    assert sig == a._fused._sig
    _edit_behind_ppos_back(a, edit); _edit_behind_ppos_back(b, edit)
    if edit != "data_write_plus_touch":
        assert a._fused.state_signature() != sig and a._fused.synced      # nobody told the object: only the signature knows
    a.train(); b.train()
    for (na, p), (_, q) in zip(a.policy.named_parameters(), b.policy.named_parameters()):
        torch.testing.assert_close(p, q, rtol=2e-3, atol=2e-5, msg=lambda m: f"{na} after {edit}: {m}")
        sa, sb = a.optimizer.state[p], b.optimizer.state[q]
        assert float(sa["step"]) == float(sb["step"])
        torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=5e-3, atol=1e-6)
        torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=5e-3, atol=1e-9)
    # ... and an untouched object still reuses its images: no reload from the module / optimiser
    reloads, load = [], a._fused.load_from_torch
    a._fused.load_from_torch = lambda: reloads.append(1) or load()
    a.train()
    assert not reloads and all(torch.isfinite(p).all() for p in a.policy.parameters())


@pytest.mark.parametrize("edit", ["param_add", "policy_load_state_dict"])
def test_edits_of_the_policy_between_two_rollouts_are_seen_by_the_collector(edit):
    """The collector's parameter image (what fw_collect_step's act waves read) follows in-place edits of the module too: the
    next rollout's log-probs are those of the EDITED policy on the observations it stored."""
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 512, seed=4)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=128, n_epochs=1, seed=4))
    ppo.collect_rollouts(); ppo.train(); ppo.collect_rollouts()
    assert ppo._flat_current
    _edit_behind_ppos_back(ppo, edit)
    ppo.collect_rollouts()                                                # (a replay of the captured graph: the image is a fixed buffer)
    torch.cuda.synchronize()
    with torch.no_grad():
        _, logp, _ = ppo.policy.evaluate_actions(ppo.buf_obs.reshape(-1, env.obs_dim), ppo.buf_act.reshape(-1, 4))
    torch.testing.assert_close(logp.reshape(ppo.buf_logp.shape), ppo.buf_logp, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(_flat_params(ppo.policy, env.obs_dim), ppo._fused.flat, rtol=0, atol=0)


def test_fused_ppo_update_rejects_what_it_cannot_run():
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    assert L.fw_ppo_param_count(28) == 2 * (28 * 64 + 64 + 64 * 64 + 64) + 64 * 4 + 4 + 64 + 1 + 4
    z = torch.zeros(64, device="cuda")
    H = R._PpoHyper()
    import ctypes as C
    args = [R._p(z)] * 8 + [R._p(z.to(torch.int32))]
    ws = torch.zeros(int(L.fw_ppo_update_workspace_bytes(1, 64, 28)), dtype=torch.uint8, device="cuda")
    assert L.fw_ppo_update_workspace_bytes(0, 64, 28) == K.FW_EINVAL and L.fw_ppo_update_workspace_bytes(1, 64, 65) == K.FW_EINVAL
    assert L.fw_ppo_update(*args, 1, 100, 28, C.byref(H), None, R._p(ws), ws.numel(), None) == K.FW_EINVAL      # batch not a multiple of 16
    assert L.fw_ppo_update(*args, 1, 64, 65, C.byref(H), None, R._p(ws), ws.numel(), None) == K.FW_EINVAL       # obs_dim too large
    assert L.fw_ppo_update(*args, 0, 64, 28, C.byref(H), None, R._p(ws), ws.numel(), None) == K.FW_EINVAL
    assert L.fw_ppo_update(*args, 1, 64, 28, C.byref(H), None, None, 0, None) == K.FW_EINVAL                    # no workspace
    assert L.fw_ppo_update(*args, 4, 64, 28, C.byref(H), None, R._p(ws), ws.numel(), None) == K.FW_EINVAL       # workspace sized for 1 minibatch
    assert b"workspace" in L.fw_last_error(None)


def _flat_params(pol, d):
    f = R.FusedPpoUpdate(pol, torch.optim.Adam(pol.parameters()), d)
    f.load_params_from_torch()
    return f.flat


@pytest.mark.parametrize("d,n,dtype", [(28, 4096, torch.float64), (56, 1000, torch.float32), (27, 65, torch.float64)])
def test_fw_policy_act_matches_the_torch_policy(d, n, dtype):
    """Deterministic mode is the torch forward to fp32 rounding; sampling mode draws unit normals, reports the matching
    log-prob, clips the env copy of the action and fills the rollout-buffer slots."""
    from pyflyt_drone_amd import _lib
    import ctypes as C
    L = _lib.lib()
    torch.manual_seed(d)
    pol = R.MlpPolicy(d).cuda()
    with torch.no_grad():
        pol.log_std.copy_(torch.tensor([-0.5, 0.0, 0.3, -1.0]))
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    flat = _flat_params(pol, d)
    obs = (torch.randn((n, d), device="cuda") * 2).clamp(-10, 10)
    rng = torch.tensor([1234, 7], dtype=torch.int64, device="cuda")
    oc, ar = torch.zeros_like(obs), torch.zeros((n, 4), device="cuda")
    ae, lp, val = torch.zeros((n, 4), device="cuda", dtype=dtype), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    def act(det, nets=3):
        _lib.check(L.fw_policy_act(R._p(flat), R._p(obs), n, d, nets, det, R._p(rng), 0, R._p(oc), R._p(ar), R._p(ae), int(dtype == torch.float64),
                                   R._p(lp), R._p(val), None))
    act(1)
    with torch.no_grad():
        a0, v0, lp0 = pol(obs, deterministic=True)
    torch.testing.assert_close(ar, a0, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(val, v0, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(lp, lp0, rtol=1e-5, atol=1e-5)
    assert torch.equal(oc, obs)
    torch.testing.assert_close(ae.float(), a0.clamp(-1, 1), rtol=1e-4, atol=2e-5)
    act(0)
    with torch.no_grad():
        z = (ar - a0) / torch.exp(pol.log_std)
        lp1 = pol._log_prob(ar, a0, pol.log_std)
    torch.testing.assert_close(lp, lp1, rtol=1e-3, atol=2e-3)
    if n >= 1000:
        assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
        assert abs(float((z[:, 0] * z[:, 1]).mean())) < 0.06                      # components are independent
    assert float(ae.abs().max()) <= 1.0
    ar_first = ar.clone()
    act(0)
    assert torch.equal(ar, ar_first)                                              # counter-based: same (seed, draw) -> same noise
    rng[1] += 1
    act(0)
    assert not torch.equal(ar, ar_first)
    before = val.clone(); val.zero_(); ar.zero_()
    act(0, nets=2)                                                                # value block only
    assert torch.equal(val, before) and float(ar.abs().max()) == 0.0


def test_fw_rollout_post_matches_vecnormalize_reward_path():
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    n = 3000
    g = torch.Generator().manual_seed(3)

    class V:
        device, num_envs, obs_dim = torch.device("cuda"), n, 5
    ref = R.VecNormalizeDevice(V(), use_fused_kernel=False)
    ret = torch.zeros(n, dtype=torch.float64, device="cuda")
    mean, var, cnt = (x.clone() for x in (ref.ret_rms.mean, ref.ret_rms.var, ref.ret_rms.count))
    rng = torch.tensor([5, 0], dtype=torch.int64, device="cuda")
    acc, want_acc = torch.zeros(3, dtype=torch.float64, device="cuda"), torch.zeros(3, dtype=torch.float64, device="cuda")
    for step in range(6):
        rew = (torch.randn(n, generator=g, dtype=torch.float64) * 30).cuda()
        term = (torch.rand(n, generator=g) < 0.1).to(torch.uint8).cuda()
        trunc = (torch.rand(n, generator=g) < 0.1).to(torch.uint8).cuda()
        tv = torch.randn(n, generator=g).cuda()
        # reference: the torch statements of VecNormalizeDevice.step + PPO._rollout_body
        ref.returns.mul_(ref.gamma).add_(rew); ref.ret_rms.update(ref.returns)
        ref_ret_before = ref.returns.clone()
        rn = (rew / torch.sqrt(ref.ret_rms.var + ref.epsilon)).clamp(-10, 10).float()
        done = (term | trunc).bool()
        rn = rn + 0.99 * tv * (trunc.bool() & ~term.bool()).float()
        ref.returns.masked_fill_(done, 0.0)
        out, start = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        want_acc += torch.stack([ref_ret_before.sum(), (ref_ret_before * ref_ret_before).sum(), torch.tensor(float(n), dtype=torch.float64, device="cuda")])
        _lib.check(L.fw_rollout_post(R._p(rew), 1, R._p(term), R._p(trunc), R._p(tv), R._p(ret), R._p(mean), R._p(var), R._p(cnt), n, 1, 1,
                                     0.99, 10.0, 1e-8, R._p(out), R._p(start), R._p(rng), R._p(acc), None))
        torch.testing.assert_close(acc, want_acc, rtol=1e-12, atol=1e-9)
        torch.testing.assert_close(out, rn, rtol=1e-5, atol=1e-5)
        assert torch.equal(start.bool(), done)
        torch.testing.assert_close(ret, ref.returns, rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(var, ref.ret_rms.var, rtol=1e-10, atol=0)
        torch.testing.assert_close(cnt, ref.ret_rms.count)
    assert int(rng[1]) == 6


def test_fused_collection_fills_the_buffers_like_the_torch_path():
    """Whole rollouts through PPO.collect_rollouts with fw_policy_act / fw_rollout_post (graph-replayed) against the
    module: stored log-probs and values are those of the stored (obs, action) pairs, episode starts follow the env's
    done flags, both normalisers see every sample, and the reward statistics match a torch-path run of the same length."""
    out = {}
    for fused in (True, False):
        env = P.FixedwingVecEnv(K.train_waypoints_v3_config(flight_dome_size=40.0), 1024, seed=9)
        ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=8, batch_size=1024, n_epochs=1, seed=9, fused_collect=fused))
        assert ppo._collect_fused == fused
        for _ in range(4):
            ppo.collect_rollouts()
        torch.cuda.synchronize()
        with torch.no_grad():
            v, lp, _ = ppo.policy.evaluate_actions(ppo.buf_obs.reshape(-1, env.obs_dim), ppo.buf_act.reshape(-1, 4))
        torch.testing.assert_close(lp.reshape(8, 1024), ppo.buf_logp, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(v.reshape(8, 1024), ppo.buf_val, rtol=1e-4, atol=1e-4)
        assert float(ppo.env.obs_rms.count) == pytest.approx(1e-4 + (4 * 8 + 1) * 1024)
        assert float(ppo.env.ret_rms.count) == pytest.approx(1e-4 + 4 * 8 * 1024)
        assert torch.isfinite(ppo.buf_rew).all() and torch.isfinite(ppo.adv).all()
        assert set(ppo.buf_start.unique().tolist()) <= {0.0, 1.0}
        out[fused] = (float(ppo.env.ret_rms.var), float(ppo.buf_start.mean()), float(ppo.buf_act.std()))
    assert out[True][0] == pytest.approx(out[False][0], rel=0.35)         # same reward scale (different action noise streams)
    assert out[True][2] == pytest.approx(out[False][2], rel=0.05)


def test_fw_policy_terminal_value_bootstraps_only_truncated_rows():
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    d, n = 28, 1000
    torch.manual_seed(4)
    pol = R.MlpPolicy(d).cuda()
    flat = _flat_params(pol, d)
    tobs = torch.randn((n, d), device="cuda", dtype=torch.float64) * 5 + 1
    mean, var = torch.randn(d, device="cuda", dtype=torch.float64), torch.rand(d, device="cuda", dtype=torch.float64) * 4 + 0.1
    term = torch.zeros(n, dtype=torch.uint8, device="cuda"); trunc = torch.zeros_like(term)
    trunc[[3, 70, 999]] = 1; term[70] = 1; trunc[500] = 1                       # row 70 is terminated as well: not a timeout
    val = torch.full((n,), -7.0, device="cuda")
    _lib.check(L.fw_policy_terminal_value(R._p(flat), R._p(tobs), 1, n, d, R._p(mean), R._p(var), 10.0, 1e-8, R._p(term), R._p(trunc), R._p(val), None))
    with torch.no_grad():
        x = ((tobs - mean) / torch.sqrt(var + 1e-8)).float().clamp(-10, 10)
        ref = pol.predict_values(x)
    for row in (3, 500, 999):
        assert float(val[row]) == pytest.approx(float(ref[row]), rel=1e-4, abs=2e-5)
    blocks_with_timeout = {3 // 64, 500 // 64, 999 // 64}
    for b in range((n + 63) // 64):
        rows = slice(b * 64, min(n, b * 64 + 64))
        if b in blocks_with_timeout:
            torch.testing.assert_close(val[rows], ref[rows], rtol=1e-4, atol=2e-5)
        else:
            assert bool((val[rows] == -7.0).all())                               # skipped blocks leave the buffer alone


def test_objlock_training_learns_to_strike():
    """The whole learner on the reference's ObjLock config and PPO hyper-parameters (train/train_objlock.py:27-86; batch 64,
    10 epochs, 32 768 samples per update): ~1.3 M steps -- a few seconds -- take the deterministic policy from never
    striking the duck to striking it in most evaluation episodes.  Everything in the loop is one of this package's kernels."""
    from pyflyt_drone_amd import evaluate
    cfg = K.train_objlock_config()
    env = R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, 4096, seed=42))
    eval_env = R.VecNormalizeDevice(P.FixedwingVecEnv(cfg, 32, seed=42, global_env_offset=4096), training=False, norm_reward=False)
    ppo = R.PPO(env, R.PPOConfig(n_steps=8, batch_size=64, n_epochs=10, learning_rate=3e-4, ent_coef=0.001, seed=42))
    assert ppo._collect_fused and R.FusedPpoUpdate.applies(ppo.policy, ppo.cfg, env.obs_dim, 64, ppo.device)
    evaluate.sync_envs_normalization(env, eval_env)
    before = evaluate.evaluate_policy(ppo.policy, eval_env, 32, deterministic=True)
    ppo.learn(1_300_000)
    evaluate.sync_envs_normalization(env, eval_env)
    after = evaluate.evaluate_policy(ppo.policy, eval_env, 32, deterministic=True)
    s0, s1 = float(np.mean(before.duck_strike)), float(np.mean(after.duck_strike))
    assert s0 <= 0.2 and s1 >= 0.5, (s0, s1)
    assert after.mean_reward > before.mean_reward + 500.0


@pytest.mark.parametrize("dtype,n,d", [(torch.float64, 4096, 28), (torch.float32, 1000, 56), (torch.float64, 65, 27)])
def test_fw_collect_stats_matches_the_torch_statistics(dtype, n, d):
    """fw_collect_stats (one launch: observation moments + discounted-return tracker + both Chan merges, last block folds in
    a fixed order) against the torch statements of VecNormalizeDevice.step, over several steps; the accumulators a sharded job
    all-reduces hold the batch sums; the action sampler's draw counter advances; flags off = statistics untouched."""
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(n + d)

    class V:
        device, num_envs, obs_dim = torch.device("cuda"), n, d
    ref = R.VecNormalizeDevice(V(), use_fused_kernel=False)
    dev = V()
    me = R.VecNormalizeDevice(dev, use_fused_kernel=True)
    oacc, racc = torch.zeros(2 * d + 1, dtype=torch.float64, device="cuda"), torch.zeros(3, dtype=torch.float64, device="cuda")
    want_o, want_r = torch.zeros_like(oacc), torch.zeros_like(racc)
    rng = torch.tensor([5, 0], dtype=torch.int64, device="cuda")
    for step in range(6):
        obs = (torch.randn((n, d), generator=g, dtype=torch.float64) * (1 + step) + 2 * step).to(dtype).cuda()
        rew = (torch.randn(n, generator=g, dtype=torch.float64) * 30).to(dtype).cuda()
        term = (torch.rand(n, generator=g) < 0.1).to(torch.uint8).cuda(); trunc = (torch.rand(n, generator=g) < 0.1).to(torch.uint8).cuda()
        upd = step != 4
        if upd:
            ref.obs_rms.update(obs)
            ref.returns.mul_(ref.gamma).add_(rew.double()); ref.ret_rms.update(ref.returns)
            x = obs.double()
            want_o += torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(n)], dtype=torch.float64, device="cuda")])
            want_r += torch.stack([ref.returns.sum(), (ref.returns ** 2).sum(), torch.tensor(float(n), dtype=torch.float64, device="cuda")])
        ref.returns.masked_fill_((term | trunc).bool(), 0.0)
        _lib.check(L.fw_collect_stats(R._p(obs), int(dtype == torch.float64), n, d, R._p(me.obs_rms.mean), R._p(me.obs_rms.var), R._p(me.obs_rms.count),
                                      int(upd), R._p(rew), int(dtype == torch.float64), R._p(term), R._p(trunc), R._p(me.returns),
                                      R._p(me.ret_rms.mean), R._p(me.ret_rms.var), R._p(me.ret_rms.count), int(upd), 0.99, R._p(rng),
                                      R._p(me._ws_stats), R._p(oacc) if upd else None, R._p(racc) if upd else None, None))
        torch.testing.assert_close(me.obs_rms.mean, ref.obs_rms.mean, rtol=1e-10, atol=1e-10)
        torch.testing.assert_close(me.obs_rms.var, ref.obs_rms.var, rtol=1e-9, atol=1e-10)
        torch.testing.assert_close(me.obs_rms.count, ref.obs_rms.count)
        torch.testing.assert_close(me.returns, ref.returns, rtol=1e-12, atol=1e-9)
        torch.testing.assert_close(me.ret_rms.var, ref.ret_rms.var, rtol=1e-10, atol=0)
        torch.testing.assert_close(me.ret_rms.mean, ref.ret_rms.mean, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(me.ret_rms.count, ref.ret_rms.count)
        torch.testing.assert_close(oacc, want_o, rtol=1e-12, atol=1e-6); torch.testing.assert_close(racc, want_r, rtol=1e-12, atol=1e-6)
    assert int(rng[1]) == 6
    assert int(me._ws_stats.view(torch.int32)[-16:].abs().sum()) == 0            # the ticket is left at zero for the next launch


def test_fw_collect_act_normalises_on_load_and_finalises_the_previous_step():
    """fw_collect_act = fw_policy_act on observations normalised on load (bit-identical to fw_normalize_obs followed by
    fw_policy_act), and its value block applies VecNormalize's reward path + SB3's truncation bootstrap to the previous step:
    checked against fw_rollout_post fed with fw_policy_terminal_value."""
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    d, n = 28, 1000
    torch.manual_seed(11)
    pol = R.MlpPolicy(d).cuda()
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    flat = _flat_params(pol, d)
    raw = torch.randn((n, d), device="cuda", dtype=torch.float64) * 4 + 1
    mean, var = torch.randn(d, device="cuda", dtype=torch.float64), torch.rand(d, device="cuda", dtype=torch.float64) * 4 + 0.1
    cnt = torch.ones(1, device="cuda", dtype=torch.float64)
    rng = torch.tensor([77, 3], dtype=torch.int64, device="cuda")
    # reference: normalise, then act
    obs_n = torch.zeros((n, d), device="cuda")
    _lib.check(L.fw_normalize_obs(R._p(raw), 1, n, d, R._p(mean), R._p(var), R._p(cnt), 0, 10.0, 1e-8, R._p(obs_n), None, None, None))
    oc0, ar0, ae0 = torch.zeros_like(obs_n), torch.zeros((n, 4), device="cuda"), torch.zeros((n, 4), device="cuda", dtype=torch.float64)
    lp0, v0 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    _lib.check(L.fw_policy_act(R._p(flat), R._p(obs_n), n, d, 3, 0, R._p(rng), 512, R._p(oc0), R._p(ar0), R._p(ae0), 1, R._p(lp0), R._p(v0), None))
    # previous step: rewards, flags, terminal observations
    g = torch.Generator().manual_seed(2)
    rew = (torch.randn(n, generator=g, dtype=torch.float64) * 20).cuda()
    term = (torch.rand(n, generator=g) < 0.05).to(torch.uint8).cuda(); trunc = (torch.rand(n, generator=g) < 0.03).to(torch.uint8).cuda()
    trunc[64:128] = 0; term[64:128] = 0                                           # one block without any episode end
    tobs = torch.randn((n, d), device="cuda", dtype=torch.float64) * 3
    ret_var = torch.tensor([7.5], dtype=torch.float64, device="cuda")
    tv = torch.zeros(n, device="cuda")
    _lib.check(L.fw_policy_terminal_value(R._p(flat), R._p(tobs), 1, n, d, R._p(mean), R._p(var), 10.0, 1e-8, R._p(term), R._p(trunc), R._p(tv), None))
    ret, rm, rc = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(1, dtype=torch.float64, device="cuda"), torch.ones(1, dtype=torch.float64, device="cuda")
    rew0, st0 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    _lib.check(L.fw_rollout_post(R._p(rew), 1, R._p(term), R._p(trunc), R._p(tv), R._p(ret), R._p(rm), R._p(ret_var.clone()), R._p(rc), n, 0, 1,
                                 0.99, 10.0, 1e-8, R._p(rew0), R._p(st0), None, None, None))
    oc1, ar1, ae1 = torch.zeros_like(obs_n), torch.zeros((n, 4), device="cuda"), torch.zeros((n, 4), device="cuda", dtype=torch.float64)
    lp1, v1 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    rew1, st1 = torch.full((n,), -9.0, device="cuda"), torch.full((n,), -9.0, device="cuda")
    _lib.check(L.fw_collect_act(R._p(flat), R._p(raw), 1, n, d, R._p(mean), R._p(var), 10.0, 1e-8, 3, 0, R._p(rng), 512, R._p(oc1), R._p(ar1),
                                R._p(ae1), 1, R._p(lp1), R._p(v1), R._p(rew), R._p(term), R._p(trunc), R._p(tobs), R._p(ret_var), 1, 10.0, 1e-8, 0.99,
                                R._p(rew1), R._p(st1), None))
    assert torch.equal(oc1, obs_n) and torch.equal(oc1, oc0)
    assert torch.equal(ar1, ar0) and torch.equal(ae1, ae0) and torch.equal(lp1, lp0) and torch.equal(v1, v0)
    assert torch.equal(st1, st0)
    timeouts = (trunc.bool() & ~term.bool())
    assert int(timeouts.sum()) >= 5
    torch.testing.assert_close(rew1, rew0, rtol=1e-6, atol=1e-6)
    assert torch.equal(rew1[~timeouts], rew0[~timeouts])
    # without a previous step nothing of it is touched; the policy block alone leaves the value outputs alone
    rew1.fill_(-9.0); v1.fill_(-9.0)
    _lib.check(L.fw_collect_act(R._p(flat), R._p(raw), 1, n, d, R._p(mean), R._p(var), 10.0, 1e-8, 1, 0, R._p(rng), 512, R._p(oc1), R._p(ar1),
                                R._p(ae1), 1, R._p(lp1), None, None, None, None, None, None, 0, 0.0, 0.0, 0.0, None, None, None))
    assert bool((rew1 == -9.0).all()) and bool((v1 == -9.0).all()) and torch.equal(ar1, ar0)
    assert L.fw_collect_act(R._p(flat), R._p(raw), 1, n, d, R._p(mean), R._p(var), 10.0, 1e-8, 1, 0, R._p(rng), 512, R._p(oc1), R._p(ar1),
                            R._p(ae1), 1, R._p(lp1), None, R._p(rew), R._p(term), R._p(trunc), R._p(tobs), R._p(ret_var), 1, 10.0, 1e-8, 0.99,
                            R._p(rew1), R._p(st1), None) == K.FW_EINVAL           # finalisation needs the value block


@pytest.mark.parametrize("task,n", [("waypoints", 4096), ("waypoints", 1000), ("waypoints", 24), ("waypoints", 8192), ("waypoints", 12288), ("objlock", 1024), ("objlock", 4096),
                                    ("combined", 520), ("combined", 2048), ("waypoints_wind", 2048), ("waypoints_wind", 8192)])
def test_one_launch_vec_step_equals_the_three_launch_collector(task, n):
    """fw_collect_step (act waves + step waves + statistics fold in ONE grid) against fw_collect_act -> fw_step ->
    fw_collect_stats on twin envs, through PPO.collect_rollouts (hipGraph replays included): the same actions, log-probs,
    values, observations, rewards and episode starts in the rollout buffers, the same normaliser statistics (the fold order
    differs: 1e-12), the env's hand-off counters agree, and no step wave ever gave up waiting for its actions."""
    cfgs = {"waypoints": lambda: K.train_waypoints_v3_config(flight_dome_size=60.0, max_duration_seconds=6.0),
            "waypoints_wind": lambda: K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND, flight_dome_size=60.0, max_duration_seconds=6.0),
            "objlock": lambda: K.train_objlock_config(max_duration_seconds=0.7), "combined": lambda: K.train_waypoint_objlock_config(max_duration_seconds=0.7)}
    runs = {}
    for one in (True, False):
        env = P.FixedwingVecEnv(cfgs[task](), n, seed=21)
        ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=8, batch_size=64, n_epochs=1, seed=3, one_launch_collect=one))
        assert ppo._collect_fused and ppo._one_launch == one
        assert env.g8_waves == (2 if (task == "waypoints" and n > 8192) or (task == "waypoints_wind" and n > 6144) else 1)      # (the two-wave build has its own collect kernel)
        bufs = []
        for _ in range(5):                                     # eager, capture, three replays
            ppo.collect_rollouts()
            bufs.append([b.clone() for b in (ppo.buf_obs, ppo.buf_act, ppo.buf_logp, ppo.buf_val, ppo.buf_rew, ppo.buf_start, ppo.last_values, ppo.last_starts,
                                             ppo.adv, ppo.ret, ppo.last_obs)])      # (one launch: adv / ret / last_obs come from fw_collect_close)
        torch.cuda.synchronize()
        st = torch.cat([ppo.env.obs_rms.mean, ppo.env.obs_rms.var, ppo.env.obs_rms.count, ppo.env.ret_rms.mean.reshape(1), ppo.env.ret_rms.var.reshape(1),
                        ppo.env.ret_rms.count, ppo.env.returns])
        status = int(ppo._ws_collect.view(torch.int32)[-16 + 3]) if one else 0        # CS_STATUS of the workspace's last 64 bytes
        runs[one] = (bufs, st, env.get_counters(), env.get_state(), int(ppo._rng[1]), status)
    (ba, sa, ca, xa, ra, status), (bb, sb, cb, xb, rb, _) = runs[True], runs[False]
    assert status == 0, "a step wave's wait for its actions ran out"
    assert ra == rb == 5 * 8
    assert ca == cb and (ca["resets"] > 0 or n < 100), (ca, cb)
    torch.testing.assert_close(sa[:-n], sb[:-n], rtol=1e-9, atol=1e-9)   # (another fold order: 1e-12 of a column's scale)
    # the per-env return trackers follow the rewards, which follow the actions: the two collectors' policy forwards (16 x 16 x 4 tiles in
    # the act waves, 32 x 32 x 2 in fw_collect_act) agree to fp32 rounding, not to the bit
    torch.testing.assert_close(sa[-n:], sb[-n:], rtol=1e-7, atol=1e-7)
    for it, (x, y) in enumerate(zip(ba, bb)):
        for name, u, v in zip(("obs", "act", "logp", "val", "rew", "start", "last_values", "last_starts", "adv", "ret", "last_obs"), x, y):
            torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-5, msg=lambda m: f"rollout {it} {name}: {m}")
        assert torch.equal(x[5], y[5]) and torch.equal(x[7], y[7])
    np.testing.assert_allclose(xa, xb, rtol=0, atol=1e-5)      # the simulators end in the same state (1e-12 in the statistics -> an ulp of a float32 observation -> 1e-7 after 40 steps)


def test_fw_collect_step_refuses_the_mappings_it_does_not_serve(monkeypatch):
    from pyflyt_drone_amd import _lib
    monkeypatch.setenv("FWSIM_LANES_PER_ENV", "1")
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 256, seed=1)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=64, n_epochs=1))
    assert ppo._collect_fused and not ppo._one_launch                  # falls back to the three-launch collector
    a = K.FwCollectArgs()
    assert _lib.lib().fw_collect_step(env._h, C.byref(a), None) == K.FW_EUNSUPPORTED
    ppo.collect_rollouts()
    assert float(ppo.env.obs_rms.count) == pytest.approx(1e-4 + 5 * 256)


def test_fw_collect_close_checks_its_arguments():
    """fw_collect_close refuses what it cannot run: the mapping without a one-launch form (FW_EUNSUPPORTED, like fw_collect_step),
    missing buffers, GAE buffers that do not match (rew_out must be row T - 1 of the rewards) -- FW_EINVAL, nothing launched."""
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 256, seed=1)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=64, n_epochs=1))
    assert ppo._one_launch and ppo._close_gae
    ppo.collect_rollouts()                                              # allocates and initialises the workspace
    a, c = K.FwCollectArgs(), K.FwCollectCloseArgs()
    assert L.fw_collect_close(env._h, C.byref(a), C.byref(c), None) == K.FW_EINVAL          # nothing filled in
    vn, venv, T = ppo.env, ppo.env.venv, 4
    a.params = ppo._fused.flat.data_ptr()
    a.obs_mean, a.obs_var, a.obs_count = vn.obs_rms.mean.data_ptr(), vn.obs_rms.var.data_ptr(), vn.obs_rms.count.data_ptr()
    a.returns = vn.returns.data_ptr()
    a.ret_mean, a.ret_var, a.ret_count = vn.ret_rms.mean.data_ptr(), vn.ret_rms.var.data_ptr(), vn.ret_rms.count.data_ptr()
    a.value, a.obs_copy = ppo.last_values.data_ptr(), ppo.last_obs.data_ptr()
    a.obs, a.reward = venv.obs.data_ptr(), venv.rewards.data_ptr()
    a.terminated, a.truncated, a.terminal_obs = venv.terminated.data_ptr(), venv.truncated.data_ptr(), venv.terminal_obs.data_ptr()
    a.workspace, a.workspace_bytes = ppo._ws_collect.data_ptr(), ppo._ws_collect.numel() * 8
    a.gamma, a.clip_obs, a.eps_obs, a.clip_reward, a.eps_reward = 0.99, 10.0, 1e-8, 10.0, 1e-8
    a.rew_out, a.start_out = ppo.buf_rew[T - 1].data_ptr(), ppo.last_starts.data_ptr()
    assert L.fw_collect_close(env._h, C.byref(a), C.byref(c), None) == K.FW_EINVAL          # no GAE buffers
    c.rewards, c.values, c.episode_starts = ppo.buf_rew.data_ptr(), ppo.buf_val.data_ptr(), ppo.buf_start.data_ptr()
    c.adv, c.ret, c.T, c.gae_gamma, c.gae_lambda = ppo._adv_buf.data_ptr(), ppo._ret_buf.data_ptr(), T, 0.99, 0.95
    a.rew_out = ppo.buf_rew[T - 2].data_ptr()
    assert L.fw_collect_close(env._h, C.byref(a), C.byref(c), None) == K.FW_EINVAL          # rew_out is not row T - 1
    a.rew_out = ppo.buf_rew[T - 1].data_ptr()
    assert L.fw_collect_close(env._h, C.byref(a), C.byref(c), None) == K.FW_OK              # complete: runs
    torch.cuda.synchronize()
    assert torch.isfinite(ppo._adv_buf).all() and torch.isfinite(ppo._ret_buf).all() and torch.isfinite(ppo.last_values).all()
    assert int(ppo._ws_collect.view(torch.int32)[-16 + 3]) == 0
