"""The in-grid hand-off protocols of the learner kernels, tested as protocols.

fw_collect_step (act waves -> step waves -> fold waves -> merge wave inside one grid) and fw_ppo_update (two / four workgroups
swapping gradients and norms through memory once per minibatch) synchronise with flags, sentinels and cache-scope tricks
instead of kernel boundaries.  A green functional suite has already hidden one real bug there (a reader that kept a stale L1
line and was right only because the line happened to be evicted).  These tests look at the protocols themselves:

* A / B bit-identity of fw_ppo_update with its exchanges on the shared-L2 path and forced onto device-scope accesses
  (FWSIM_PPO_NO_L2_SWAP=1): the arithmetic is the same, only the coherence mechanism differs, so after >= 10 240 sequential
  minibatches ANY stale read shows as a different bit somewhere in the parameters or Adam moments;
* provoked timeouts (FWSIM_SPIN_LOG2 shrinks every bounded wait to one poll): the kernels must still drain, and the product
  must raise on the host side -- SB3's contract for the reference's SubprocVecEnv is "step returns or raises"
  (train/train_Fixedwing_Waypoints_v3.py:251), never silent garbage;
* the collector above 1024 step workgroups (every partial-sum slot folded), where the previous fold silently dropped slots.
"""
import math

import numpy as np
import pytest
import torch

import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K
from pyflyt_drone_amd import rollout as R
from test_rollout_gpu import _filled_ppo

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.detach().contiguous().view(torch.int32)


@pytest.mark.parametrize("d,bs,split", [(28, 128, None), (56, 64, None), (28, 192, None), (27, 256, None), (28, 128, "64x2"), (28, 256, "32x4"),
                                         (28, 128, "32x4"), (40, 512, None), (56, 128, None), (28, 128, "all-to-all"), (40, 512, "all-to-all")])
def test_every_block_of_a_network_ends_the_update_with_the_same_bits(d, bs, split, monkeypatch):
    """The blocks of a network (up to eight) each apply Adam to their own LDS copy of the weights and their own register copy of the
    moments, from gradient sums they each form themselves out of the same four partials: the sums must be formed in the same order
    everywhere, or the copies drift apart.  FWSIM_PPO_WRITER=last makes the last block write the result back instead of the first."""
    T, n = (3, 256) if bs == 192 else (4, 256)
    if split == "all-to-all":                       # (default with four blocks: reduce-scatter of the gradient tiles + all-gather of the weights)
        monkeypatch.setenv("FWSIM_PPO_RS", "0")
    elif split is not None:
        monkeypatch.setenv("FWSIM_PPO_SPLIT", split)
    runs = {}
    for last in (False, True):
        if last:
            monkeypatch.setenv("FWSIM_PPO_WRITER", "last")
        else:
            monkeypatch.delenv("FWSIM_PPO_WRITER", raising=False)
        ppo = _filled_ppo(True, d, bs, 64, T=T, n=n, seed=5)
        ppo.train()
        torch.cuda.synchronize()
        f = ppo._fused
        runs[last] = (f.flat.clone(), f.mom_m[f._slot].clone(), f.mom_v[f._slot].clone())
    for x, y in zip(runs[False], runs[True]):
        assert torch.isfinite(x).all()
        assert torch.equal(_bits(x), _bits(y)), "the first and the last block of a network hold different bits"


@pytest.mark.parametrize("d,bs,form", [(28, 128, "rs"), (56, 64, "rs"), (27, 256, "rs"), (28, 128, "all-to-all"), (28, 128, "32x4")])
def test_ppo_update_is_bit_identical_on_the_shared_l2_and_the_device_scope_path(d, bs, form, monkeypatch):
    """("rs" = the default cut: eight blocks per network for 128 / 256 samples, four for 64; "32x4": round 4's cut of 128 samples)"""
    if form == "all-to-all":
        monkeypatch.setenv("FWSIM_PPO_RS", "0")
    elif form != "rs":
        monkeypatch.setenv("FWSIM_PPO_SPLIT", form)
    T, n = 4, 256
    n_epochs = math.ceil(10240 / (T * n // bs))                      # >= 10 240 sequential minibatches
    runs = {}
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("FWSIM_PPO_NO_L2_SWAP", "1")
        else:
            monkeypatch.delenv("FWSIM_PPO_NO_L2_SWAP", raising=False)
        ppo = _filled_ppo(True, d, bs, n_epochs, T=T, n=n, seed=11)
        ppo.train()
        torch.cuda.synchronize()
        f = ppo._fused
        owned = f._slot
        runs[forced] = (f.flat.clone(), f.mom_m[owned].clone(), f.mom_v[owned].clone(), f.last_paths,
                        [p.detach().clone() for p in ppo.policy.parameters()], dict(ppo.logs))
    (pa, ma, va, paths_a, wa, la), (pb, mb, vb, paths_b, wb, lb) = runs[False], runs[True]
    assert paths_b == 0, "FWSIM_PPO_NO_L2_SWAP=1 must force every exchange onto device-scope accesses"
    if paths_a == 0:
        pytest.skip("the working blocks were not placed on one XCD on this device: both runs took the device-scope path")
    if bs >= 64:
        assert paths_a & 0x5555, f"no gradient swap went through the shared L2 (paths {paths_a:#x})"
    assert torch.isfinite(pa).all() and torch.isfinite(ma).all() and torch.isfinite(va).all()
    assert torch.equal(_bits(pa), _bits(pb)), "parameters differ between the shared-L2 and the device-scope hand-off: a stale read"
    assert torch.equal(_bits(ma), _bits(mb)) and torch.equal(_bits(va), _bits(vb)), "Adam moments differ between the two hand-offs"
    for x, y in zip(wa, wb):
        assert torch.equal(_bits(x), _bits(y))
    assert la == lb


def test_ppo_update_wait_that_runs_out_raises_and_leaves_the_policy_alone(monkeypatch):
    """One poll per wait: some workgroup's partner has not published yet -> status word, early exit of every workgroup, a
    RuntimeError on the host side, module and optimiser untouched; the next (healthy) call works."""
    ppo = _filled_ppo(True, 28, 128, 2, seed=3)
    before = [p.detach().clone() for p in ppo.policy.parameters()]
    monkeypatch.setenv("FWSIM_SPIN_LOG2", "0")
    with pytest.raises(RuntimeError, match="fw_ppo_update gave up"):
        ppo.train()
    torch.cuda.synchronize()
    for p, q in zip(ppo.policy.parameters(), before):
        assert torch.equal(p, q)
    assert not ppo.optimizer.state or all(float(st["step"]) == 0 for st in ppo.optimizer.state.values())
    monkeypatch.delenv("FWSIM_SPIN_LOG2")
    ppo.train()
    torch.cuda.synchronize()
    assert any(not torch.equal(p, q) for p, q in zip(ppo.policy.parameters(), before))
    assert all(torch.isfinite(p).all() for p in ppo.policy.parameters())
    # ... and equals a twin that never saw the failed call (same buffers, permutation stream advanced identically)
    twin = _filled_ppo(True, 28, 128, 2, seed=3)
    for _ in range(2):                                                    # (the failed call drew its two permutations first)
        torch.randperm(4 * 256, device="cuda", generator=twin.perm_gen)
    twin.train()
    for p, q in zip(ppo.policy.parameters(), twin.policy.parameters()):
        torch.testing.assert_close(p, q, rtol=0, atol=0)


def test_collect_step_wait_that_runs_out_raises_in_python(monkeypatch):
    """PPOConfig.collect_fallback=False: SB3's "step returns or raises", and the object stays on the one-launch collector."""
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 1024, seed=5)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=128, n_epochs=1, seed=5, collect_fallback=False))
    assert ppo._one_launch
    ppo.collect_rollouts()
    ppo.check_collect_status()                                            # a healthy rollout: nothing raised
    monkeypatch.setenv("FWSIM_SPIN_LOG2", "0")                            # every bounded wait: one poll
    ppo.invalidate_graphs()                                               # (the budget is a kernel argument: not from a captured graph)
    ppo.collect_rollouts()
    with pytest.raises(RuntimeError, match="fw_collect_step: status word"):
        ppo.train()                                                       # the update refuses a void rollout
    monkeypatch.delenv("FWSIM_SPIN_LOG2")
    # the object is usable again: fresh workspace, clean status, graph re-captured on the way
    for _ in range(3):
        ppo.collect_rollouts()
    ppo.check_collect_status()
    ppo.train()
    torch.cuda.synchronize()
    assert int(ppo._ws_collect.view(torch.int32)[-16 + 3]) == 0
    assert torch.isfinite(ppo.buf_obs).all() and torch.isfinite(ppo.adv).all()
    assert ppo._one_launch and ppo.collect_fallbacks == 0


@pytest.mark.parametrize("via", ["train", "learn", "next_rollout"])
def test_a_void_rollout_is_taken_back_and_training_goes_on_on_the_three_launch_collector(via, monkeypatch):
    """Default (PPOConfig.collect_fallback): the one-launch collector's forward progress rests on in-order workgroup dispatch,
    which the platform does not promise.  Where the order differs every rollout would end in a status error -- so the first one
    moves the SAME object (same process, other code path: fw_collect_act -> fw_step -> fw_collect_stats, no in-grid wait) and
    training continues: the void rollout is collected again, the normalisers never see it, the timestep counter counts it once."""
    T, N = 4, 1024
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), N, seed=5)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=T, batch_size=128, n_epochs=1, seed=5))
    assert ppo._one_launch and ppo._close_gae
    ppo.collect_rollouts(); ppo.train()                                   # rollout 0: healthy
    torch.cuda.synchronize()
    c_obs, c_ret, steps = float(ppo.env.obs_rms.count), float(ppo.env.ret_rms.count), ppo.num_timesteps
    assert c_obs == pytest.approx(1e-4 + (T + 1) * N) and steps == T * N
    before = [p.detach().clone() for p in ppo.policy.parameters()]
    monkeypatch.setenv("FWSIM_SPIN_LOG2", "0")                            # rollout 1: every bounded wait gives up after one poll
    ppo.invalidate_graphs()
    collect = ppo.collect_rollouts

    def collect_then_restore_the_budget():                                # (the budget is read per launch; fw_ppo_update reads it too and must get its own)
        collect()
        monkeypatch.delenv("FWSIM_SPIN_LOG2", raising=False)
    ppo.collect_rollouts = collect_then_restore_the_budget
    with pytest.warns(RuntimeWarning, match="re-armed on the three-launch collector"):
        if via == "train":
            ppo.collect_rollouts()
            ppo.train()                                                   # sees the word, takes the rollout back, collects it again, updates
        elif via == "learn":
            ppo.learn(T * N, reset_num_timesteps=False)                   # one rollout + one update: never raises
        else:
            ppo.collect_rollouts()
            ppo.collect_rollouts()                                        # (a caller that only collects: the next rollout notices and replaces it)
            ppo.train()
    torch.cuda.synchronize()
    assert ppo.collect_fallbacks == 1 and not ppo._one_launch and not ppo._close_gae and ppo._collect_fused
    # exactly ONE more rollout in the statistics and in the counter: the void one left nothing behind
    assert float(ppo.env.obs_rms.count) == pytest.approx(c_obs + T * N) and float(ppo.env.ret_rms.count) == pytest.approx(c_ret + T * N)
    assert ppo.num_timesteps == steps + T * N
    assert torch.isfinite(ppo.buf_obs).all() and torch.isfinite(ppo.adv).all() and (ppo.buf_act.abs().sum(-1) > 0).all()      # (a void rollout flies zero actions)
    assert any(not torch.equal(p, q) for p, q in zip(ppo.policy.parameters(), before))
    # rollout k + 1 and on: the three-launch collector, whatever the spin budget (it has no in-grid wait), graph replays included
    for _ in range(3):
        ppo.collect_rollouts()
    ppo.check_collect_status()
    ppo.train()
    torch.cuda.synchronize()
    assert float(ppo.env.obs_rms.count) == pytest.approx(c_obs + 4 * T * N) and ppo.num_timesteps == steps + 4 * T * N
    assert ppo.collect_fallbacks == 1 and all(torch.isfinite(p).all() for p in ppo.policy.parameters())


def test_a_void_rollout_leaves_no_advantages_for_train_to_pick_up(monkeypatch):
    """a caller that catches the error of check_collect_status() and calls train() again must not train on void data."""
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 512, seed=6)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=128, n_epochs=1, seed=6, collect_fallback=False))
    ppo.collect_rollouts(); ppo.check_collect_status()
    c_obs = float(ppo.env.obs_rms.count)
    rets = ppo.env.returns.clone()
    monkeypatch.setenv("FWSIM_SPIN_LOG2", "0")
    ppo.invalidate_graphs()
    ppo.collect_rollouts()
    with pytest.raises(RuntimeError, match="fw_collect_step: status word"):
        ppo.check_collect_status()
    assert ppo.adv is None and ppo.ret is None
    assert float(ppo.env.obs_rms.count) == c_obs and torch.equal(ppo.env.returns, rets)      # statistics and return trackers of before the void rollout
    with pytest.raises(RuntimeError, match="train\\(\\) without a rollout"):
        ppo.train()
    from pyflyt_drone_amd import checkpoint
    monkeypatch.delenv("FWSIM_SPIN_LOG2")
    ppo.collect_rollouts()
    checkpoint.snapshot(ppo, include_env_state=False)                     # (looks at the word too: a healthy rollout passes)


def test_collect_step_status_through_the_c_abi():
    import ctypes as C
    from pyflyt_drone_amd import _lib
    L = _lib.lib()
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 256, seed=1)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=64, n_epochs=1))
    ppo.collect_rollouts()
    st = C.c_uint32(99)
    assert L.fw_collect_status(env._h, ppo._ws_collect.data_ptr(), ppo._ws_collect.numel() * 8, C.byref(st), None) == K.FW_OK
    assert st.value == 0
    assert L.fw_collect_status(env._h, ppo._ws_collect.data_ptr(), 64, C.byref(st), None) == K.FW_EINVAL
    # a workspace that never went through fw_collect_workspace_init is reported, not trusted
    ws = torch.zeros_like(ppo._ws_collect)
    a = K.FwCollectArgs()
    vn, venv = ppo.env, ppo.env.venv
    a.params = ppo._fused.flat.data_ptr()
    a.obs_mean, a.obs_var, a.obs_count = vn.obs_rms.mean.data_ptr(), vn.obs_rms.var.data_ptr(), vn.obs_rms.count.data_ptr()
    a.returns = vn.returns.data_ptr()
    a.ret_mean, a.ret_var, a.ret_count = vn.ret_rms.mean.data_ptr(), vn.ret_rms.var.data_ptr(), vn.ret_rms.count.data_ptr()
    a.rng = ppo._rng.data_ptr()
    a.obs_copy, a.act_raw, a.logp, a.value = ppo.buf_obs[0].data_ptr(), ppo.buf_act[0].data_ptr(), ppo.buf_logp[0].data_ptr(), ppo.buf_val[0].data_ptr()
    a.act_env = ppo._act_env.data_ptr()
    a.obs, a.reward = venv.obs.data_ptr(), venv.rewards.data_ptr()
    a.terminated, a.truncated = venv.terminated.data_ptr(), venv.truncated.data_ptr()
    a.terminal_obs, a.info_i32 = venv.terminal_obs.data_ptr(), venv.info.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
    a.gamma, a.clip_obs, a.eps_obs, a.clip_reward, a.eps_reward = 0.99, 10.0, 1e-8, 10.0, 1e-8
    a.update_obs, a.update_ret, a.norm_reward = 1, 1, 1
    assert L.fw_collect_step(env._h, C.byref(a), None) == K.FW_OK         # (zeroed slots read as "arrived": nothing waits long)
    assert L.fw_collect_status(env._h, ws.data_ptr(), ws.numel() * 8, C.byref(st), None) == K.FW_OK
    assert st.value & 8, st.value


def test_a_policy_that_produces_nan_actions_is_reported_not_flown_at_minus_one():
    env = P.FixedwingVecEnv(K.train_waypoints_v3_config(), 512, seed=2)
    ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=128, n_epochs=1, seed=2))
    ppo.collect_rollouts()
    ppo.check_collect_status()
    with torch.no_grad():
        ppo.policy.log_std.fill_(float("nan"))
    ppo._flat_current = False
    ppo.invalidate_graphs()
    ppo.collect_rollouts()
    with pytest.raises(RuntimeError, match="NaN action"):
        ppo.check_collect_status()


@pytest.mark.parametrize("task,n", [("combined", 16384), ("objlock", 12288)])
def test_one_launch_collector_folds_every_slot_above_1024_step_workgroups(task, n):
    """the fold waves used to sum at most 1024 partial-sum slots (8192 envs) while fw_create keeps the camera
    tasks on the 8-lane one-wave mapping up to 16 384 envs -- statistics silently covered part of the envs.  Against the
    three-launch collector on twin envs: same statistics, same sample counts, status 0."""
    cfgs = {"objlock": lambda: K.train_objlock_config(max_duration_seconds=0.7), "combined": lambda: K.train_waypoint_objlock_config(max_duration_seconds=0.7)}
    runs = {}
    for one in (True, False):
        env = P.FixedwingVecEnv(cfgs[task](), n, seed=21)
        assert env.lanes_per_env == 8
        ppo = R.PPO(R.VecNormalizeDevice(env), R.PPOConfig(n_steps=4, batch_size=64, n_epochs=1, seed=3, one_launch_collect=one))
        assert ppo._collect_fused and ppo._one_launch == one
        for _ in range(3):                                     # eager, capture, replay
            ppo.collect_rollouts()
        ppo.check_collect_status()
        torch.cuda.synchronize()
        vn = ppo.env
        runs[one] = (torch.cat([vn.obs_rms.mean, vn.obs_rms.var, vn.obs_rms.count, vn.ret_rms.mean.reshape(1), vn.ret_rms.var.reshape(1), vn.ret_rms.count]),
                     ppo.buf_obs.clone(), ppo.buf_rew.clone(), ppo.adv.clone(), env.get_counters())
    (sa, oa, ra, aa, ca), (sb, ob, rb, ab, cb) = runs[True], runs[False]
    assert float(sa[-1]) == pytest.approx(1e-4 + 3 * 4 * n) and float(sb[-1]) == pytest.approx(float(sa[-1]))
    torch.testing.assert_close(sa, sb, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ra, rb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(aa, ab, rtol=1e-4, atol=1e-4)
    assert ca == cb
