"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
with the CPU oracle).  CPU leg: the oracle still reproduces them.  GPU leg: the
HIP path reproduces them through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {name: (cfg, n, steps, seed) for name, cfg, kind, n, steps, seed in make_golden.cases()}


def _replay(env_step, reset, g, atol):
    np.testing.assert_allclose(reset(), g["obs0"], rtol=0, atol=atol)
    for t in range(g["actions"].shape[0]):
        o, r, te, tr, to, info = env_step(g["actions"][t])
        np.testing.assert_array_equal(te, g["terminated"][t], err_msg=f"terminated @ {t}")
        np.testing.assert_array_equal(tr, g["truncated"][t], err_msg=f"truncated @ {t}")
        np.testing.assert_array_equal(info, g["info"][t], err_msg=f"info @ {t}")
        np.testing.assert_allclose(o, g["obs"][t], rtol=0, atol=atol, err_msg=f"obs @ {t}")
        np.testing.assert_allclose(r, g["reward"][t], rtol=0, atol=atol, err_msg=f"reward @ {t}")
        d = (te | tr).astype(bool)
        np.testing.assert_allclose(to[d], g["terminal_obs"][t][d], rtol=0, atol=atol, err_msg=f"terminal_obs @ {t}")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle, name):
    cfg, n, steps, seed = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    env = oracle.OracleEnv(cfg, n, seed=seed)
    _replay(env.step, env.reset, g, atol=1e-11)
    np.testing.assert_allclose(env.get_state(), g["final_state"], rtol=0, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_golden(name):
    import torch
    import pyflyt_drone_amd as P
    cfg, n, steps, seed = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    env = P.FixedwingVecEnv(cfg, n, seed=seed)

    def step(a):
        env.step_tensor(torch.as_tensor(a, device=env.device))
        return (env.obs.cpu().numpy(), env.rewards.cpu().numpy(), env.terminated.cpu().numpy(),
                env.truncated.cpu().numpy(), env.terminal_obs.cpu().numpy(), env.info.cpu().numpy())

    _replay(step, lambda: env.reset_tensor().cpu().numpy(), g, atol=1e-7)
    np.testing.assert_allclose(env.get_state(), g["final_state"], rtol=0, atol=1e-7)
