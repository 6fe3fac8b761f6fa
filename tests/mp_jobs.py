"""Jobs run by the two helper processes that tests/conftest.py starts BEFORE the pytest process touches the GPU (a
process that has initialised the GPU must not start other programs on the GPU boxes; these helpers are started early
and idle until a test hands them a job).  Each job is one rank of a world_size-2 ``gloo`` job whose ranks share cuda:0
(RCCL refuses two ranks on one device; the product code stages collectives through the host for non-RCCL backends)."""
import os
import traceback


def serve(rank, world, job_q, res_q):
    while True:
        job = job_q.get()
        if job is None:
            return
        name, port, kwargs = job
        try:
            import torch.distributed as td
            os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
            td.init_process_group("gloo", rank=rank, world_size=world)
            try:
                out = globals()[name](rank, world, **kwargs)
            finally:
                td.destroy_process_group()
            res_q.put((rank, "ok", out))
        except BaseException:                                   # report, keep serving
            res_q.put((rank, "error", traceback.format_exc()))


def launch_serve(job_q, res_q):
    """Third helper: never touches the GPU itself, so it may start other programs for the whole session (e.g. `bench.py
    --gpus 2`, which launches its own ranks).  Jobs are (argv, extra env, timeout); results (returncode, stdout, stderr)."""
    import subprocess
    while True:
        job = job_q.get()
        if job is None:
            return
        argv, env, timeout = job
        try:
            e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            e.update(env)
            p = subprocess.run(argv, env=e, capture_output=True, text=True, timeout=timeout)
            res_q.put(("ok", (p.returncode, p.stdout, p.stderr)))
        except BaseException:
            res_q.put(("error", traceback.format_exc()))


def sharded_training(rank, world, task, n_envs, n_steps, batch_size, n_epochs, iterations, dist_update="replicated"):
    """One rank of a sharded training job: env shard at global_env_offset = rank * n_envs, fused collector (hipGraph),
    rollout all-gather, fused replicated update."""
    import numpy as np
    import torch
    import pyflyt_drone_amd as P
    from pyflyt_drone_amd import config as K
    from pyflyt_drone_amd import rollout as R
    torch.cuda.set_device(0)
    cfg = {"waypoints": K.train_waypoints_v3_config, "combined": K.train_waypoint_objlock_config}[task]()
    venv = P.FixedwingVecEnv(cfg, n_envs, device=0, seed=42, global_env_offset=rank * n_envs)
    env = R.VecNormalizeDevice(venv)
    ppo = R.PPO(env, R.PPOConfig(n_steps=n_steps, batch_size=batch_size, n_epochs=n_epochs, seed=42, dist_update=dist_update))
    facts = dict(fused_collect=bool(ppo._collect_fused), graphs=bool(ppo._graphs), replicated=bool(ppo._replicated),
                 stats_sync=env.stats_sync, use_fused_norm=bool(env.use_fused))
    first_obs = None
    checks = []
    for it in range(iterations):
        ppo.collect_rollouts()
        if first_obs is None:
            first_obs = ppo.buf_obs[0].clone().cpu().numpy()
        ppo.train()
        checks.append(ppo.replica_checksum())
    facts["fused_update"] = ppo._fused is not None and ppo._flat_current
    facts["graph_captured"] = ppo._g_rollout is not None
    flat = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).cpu().numpy()
    stats = torch.cat([env.obs_rms.mean, env.obs_rms.var, env.obs_rms.count, env.ret_rms.var.reshape(1), env.ret_rms.count]).cpu().numpy()
    ctr = venv.get_counters()
    return dict(facts=facts, checks=checks, weights=flat, stats=stats, counters=ctr, num_timesteps=ppo.num_timesteps,
                allgather_bytes=ppo.allgather_bytes, allgather_ms=ppo.allgather_ms, first_obs=first_obs,
                logs=dict(ppo.logs), finite=bool(np.isfinite(flat).all()))


def status_agreement(rank, world, n_envs, fallback=False):
    """Both ranks collect a clean rollout; rank 1's copy of the collector status word is then poisoned (as if a wait inside one of
    ITS launches had run out).  train() is the collective point: both ranks must raise, rank 0 on rank 1's word."""
    import torch
    import pyflyt_drone_amd as P
    from pyflyt_drone_amd import config as K
    from pyflyt_drone_amd import rollout as R
    torch.cuda.set_device(0)
    venv = P.FixedwingVecEnv(K.train_waypoints_v3_config(), n_envs, device=0, seed=42, global_env_offset=rank * n_envs)
    ppo = R.PPO(R.VecNormalizeDevice(venv), R.PPOConfig(n_steps=4, batch_size=128, n_epochs=1, seed=42, collect_fallback=bool(fallback)))
    assert ppo._one_launch
    ppo.collect_rollouts()
    torch.cuda.synchronize()
    assert int(ppo._status_host.item()) == 0
    if rank == 1:
        ppo._status_host.fill_(2)                     # "a fold wave summed without every partial sum"
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        try:
            ppo.train()
        except RuntimeError as e:
            msg = str(e)
        else:
            msg = "no error"
    warned = [str(w.message) for w in caught if "three-launch collector" in str(w.message)]
    after = dict(one_launch=bool(ppo._one_launch), fallbacks=int(ppo.collect_fallbacks), warned=warned, num_timesteps=int(ppo.num_timesteps),
                 obs_count=float(ppo.env.obs_rms.count), checksum=ppo.replica_checksum())
    # the object is usable afterwards on both ranks
    ppo.collect_rollouts(); ppo.train()
    # ... and the same for the update: rank 1's fw_ppo_update is made to give up (one poll per wait); rank 0's runs to its end
    import os
    ppo.collect_rollouts()
    torch.cuda.synchronize()
    if rank == 1:
        os.environ["FWSIM_SPIN_LOG2"] = "0"
    try:
        ppo.train()
    except RuntimeError as e:
        msg2 = str(e)
    else:
        msg2 = "no error"
    finally:
        os.environ.pop("FWSIM_SPIN_LOG2", None)
    return dict(msg=msg, msg2=msg2, checksum=ppo.replica_checksum(), after=after)
