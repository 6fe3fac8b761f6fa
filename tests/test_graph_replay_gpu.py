"""What bench.py times is what the parity tests check: its ``Stepper`` (a captured hipGraph of 64 fw_step launches over
the pool of 64 action tensors) replayed at BASELINE.json's size against the CPU oracle, for the headline task and the
three other step kernels.

The env.step semantics of envs/fixedwing_envs/fixedwing_base_env.py:314-348 (and the SB3 worker's auto-reset) must
survive replay: the launch index the shadow / scenario hand-off keys on lives in device memory (fwsim_device.hpp:
launch_index), so a replayed node sees the same index an eager launch would -- the replayed env is compared with the
oracle after every replay AND bit for bit with a twin env stepped eagerly, and the exported counters prove that the
resets really took the hand-off, not only its in-kernel fallback.
"""
import numpy as np
import pytest

import pyflyt_drone_amd as P
from pyflyt_drone_amd import config as K

pytestmark = pytest.mark.gpu


def _cfg(which):
    if which == "waypoints":
        return K.train_waypoints_v3_config()                                  # the bench.py workload (configs[1])
    if which == "waypoints_gust":
        return K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND)   # wind acts on the dynamics: shadow warm-up
    if which == "objlock":
        return K.train_objlock_config()                                        # configs[2]'s env
    return K.train_waypoint_objlock_config()                                   # configs[4]'s env


def _compare(env, ora_out, ora, obj, tag):
    o_obs, o_rew, o_term, o_trunc, o_tobs, o_info = ora_out
    atol = 2e-5 if obj else 1e-7            # ObjLock observations are float32-rounded (flatten_objlock_env.py:46)
    assert np.array_equal(env.terminated.cpu().numpy(), o_term), tag
    assert np.array_equal(env.truncated.cpu().numpy(), o_trunc), tag
    assert np.array_equal(env.info.cpu().numpy(), o_info), tag
    np.testing.assert_allclose(env.obs.cpu().numpy(), o_obs, rtol=0, atol=atol, err_msg=f"obs {tag}")
    np.testing.assert_allclose(env.rewards.cpu().numpy(), o_rew, rtol=0, atol=1e-7, err_msg=f"reward {tag}")
    done = (o_term | o_trunc).astype(bool)
    if done.any():
        np.testing.assert_allclose(env.terminal_obs.cpu().numpy()[done], o_tobs[done], rtol=0, atol=atol, err_msg=f"tobs {tag}")
    np.testing.assert_allclose(env.get_state(), ora.get_state(), rtol=0, atol=atol, err_msg=f"state {tag}")


@pytest.mark.parametrize("which", ["waypoints", "waypoints_gust", "objlock", "combined"])
def test_bench_stepper_graph_replay_matches_oracle(oracle, which):
    import torch
    import bench
    cfg = _cfg(which)
    n, replays = 4096, 4
    obj = cfg.task == K.FW_TASK_OBJLOCK
    env = P.FixedwingVecEnv(cfg, n, seed=42)
    twin = P.FixedwingVecEnv(cfg, n, seed=42)
    assert env._h is not None and oracle.set_threads(0) >= 1
    ora = oracle.OracleEnv(cfg, n, seed=42)
    np.testing.assert_allclose(env.reset_tensor().cpu().numpy(), ora.reset(), rtol=0, atol=1e-9)
    twin.reset_tensor()
    pool = bench.action_pool(n, env.torch_dtype, env.device)
    stepper = bench.Stepper(env, pool, use_graph=True, graph_len=bench.POOL)
    assert stepper.graph is not None and stepper.graph_len == 64
    for a in pool[:2]:                       # the two eager launches Stepper issues before capturing
        ora.step(a.cpu().numpy()); twin.step_tensor(a)
    ends = 0
    for r in range(replays):
        acts = stepper.actions_of(bench.POOL)
        stepper.run(bench.POOL)
        out = None
        for a in acts:
            out = ora.step(a.cpu().numpy())
            ends += int((out[2] | out[3]).sum())
            twin.step_tensor(a)
        torch.cuda.synchronize()
        _compare(env, out, ora, obj, f"{which} replay {r}")
        # replayed == eager, bit for bit (same launch numbering => same resets take the hand-off)
        assert torch.equal(env.obs, twin.obs) and torch.equal(env.rewards, twin.rewards), r
        assert np.array_equal(env.get_state(), twin.get_state()), r
    assert stepper.counts == {"replays": replays, "eager": 0}
    c, ct = env.get_counters(), twin.get_counters()
    assert c == ct, (c, ct)
    assert c["launches"] == 2 + replays * bench.POOL
    assert c["resets"] == ends and ends > 50, (c, ends)
    hits = c["scenario_hits"] if which == "waypoints" else c["shadow_hits"]
    assert hits + c["fallbacks"] == c["resets"]
    assert hits >= 0.5 * c["resets"], f"the hand-off was meant to serve most resets under replay: {c}"


def test_single_launch_graph_still_takes_the_hand_off():
    """A graph holding ONE fw_step (an n_steps=1 collector, a user-captured step): every replay is a new launch index,
    so shadows are still built and swapped in, and the result equals eager stepping bit for bit."""
    import torch
    import bench
    cfg = K.train_waypoints_v3_config(wind_config=K.TRAIN_OBJLOCK_WIND, flight_dome_size=40.0)
    n = 1024
    env, twin = P.FixedwingVecEnv(cfg, n, seed=3), P.FixedwingVecEnv(cfg, n, seed=3)
    env.reset_tensor(); twin.reset_tensor()
    pool = bench.action_pool(n, env.torch_dtype, env.device)[:1]
    stepper = bench.Stepper(env, pool, use_graph=True, graph_len=1)
    for a in pool[:2]:
        twin.step_tensor(a)
    stepper.run(200)
    for _ in range(200):
        twin.step_tensor(pool[0])
    torch.cuda.synchronize()
    assert torch.equal(env.obs, twin.obs) and np.array_equal(env.get_state(), twin.get_state())
    c = env.get_counters()
    assert c == twin.get_counters() and c["launches"] == 201
    assert c["resets"] > 100 and c["shadow_hits"] >= 0.5 * c["resets"], c
