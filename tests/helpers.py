"""Shared helpers for the parity tests (HIP path vs CPU oracle on identical inputs)."""
import numpy as np


def seeded_actions(rng, n, kind="uniform"):
    a = rng.uniform(-1.0, 1.0, size=(n, 4))
    if kind == "gentle":           # keeps most aircraft flying for a long time
        a[:, :3] *= 0.15
        a[:, 3] = rng.uniform(-0.2, 0.6, size=n)
    return a


def run_lockstep(hip_env, ora_env, steps, rng, kind="uniform", atol=1e-6, rtol=1e-9, check_state=True, state_atol=None):
    """Step both implementations with the same actions; assert per-step agreement.
    Returns a dict of the worst deviations seen."""
    import torch
    n = hip_env.num_envs
    worst = dict(obs=0.0, rew=0.0, tobs=0.0, state=0.0, dones=0, resets=0)
    obs_h = hip_env.reset_tensor().cpu().numpy()
    obs_o = ora_env.reset()
    np.testing.assert_allclose(obs_h, obs_o, rtol=rtol, atol=atol)
    for t in range(steps):
        a = seeded_actions(rng, n, kind).astype(hip_env.np_dtype)
        o_obs, o_rew, o_term, o_trunc, o_tobs, o_info = ora_env.step(a)
        hip_env.step_tensor(torch.as_tensor(a, device=hip_env.device))
        h_obs = hip_env.obs.cpu().numpy(); h_rew = hip_env.rewards.cpu().numpy()
        h_term = hip_env.terminated.cpu().numpy(); h_trunc = hip_env.truncated.cpu().numpy()
        h_tobs = hip_env.terminal_obs.cpu().numpy(); h_info = hip_env.info.cpu().numpy()
        assert np.array_equal(h_term, o_term), f"terminated differs at step {t}: {np.nonzero(h_term != o_term)[0][:8]}"
        assert np.array_equal(h_trunc, o_trunc), f"truncated differs at step {t}"
        assert np.array_equal(h_info, o_info), f"info differs at step {t}: rows {np.nonzero((h_info != o_info).any(1))[0][:8]}"
        np.testing.assert_allclose(h_obs, o_obs, rtol=rtol, atol=atol, err_msg=f"obs step {t}")
        np.testing.assert_allclose(h_rew, o_rew, rtol=rtol, atol=atol, err_msg=f"reward step {t}")
        done = (o_term | o_trunc).astype(bool)
        if done.any() and hip_env.cfg.auto_reset:
            np.testing.assert_allclose(h_tobs[done], o_tobs[done], rtol=rtol, atol=atol, err_msg=f"terminal_obs step {t}")
            worst["tobs"] = max(worst["tobs"], float(np.abs(h_tobs[done] - o_tobs[done]).max()))
        worst["obs"] = max(worst["obs"], float(np.abs(h_obs - o_obs).max()))
        worst["rew"] = max(worst["rew"], float(np.abs(h_rew - o_rew).max()))
        worst["dones"] += int(done.sum())
    if check_state:
        sh, so = hip_env.get_state(), ora_env.get_state()
        np.testing.assert_allclose(sh, so, rtol=rtol, atol=atol if state_atol is None else state_atol, err_msg="final canonical state")
        worst["state"] = float(np.abs(sh - so).max())
    return worst
