"""The multi-GPU half of north_star on the hardware that is available: the per-GPU share of BASELINE.json's configs[3]
(4096 waypoint envs per rank at global_env_offset = rank * 4096) and configs[4] (2048 combined waypoint -> duck envs per
rank) as a world_size-2 job -- fused hipGraph collector on every rank, one all-gather of the rollout shard at update time,
the same fused fw_ppo_update on every rank.  (Two ranks share the one GPU of the box, so the collectives run over gloo;
with RCCL the same code path issues all_gather_into_tensor / all_reduce on device tensors.)
Reference callers: train/train_Fixedwing_Waypoints_v3.py:293-337, train/train_Fixedwing_Waypoints_ObjLock.py:287-403."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task,n_envs,n_steps,batch", [("waypoints", 4096, 4, 128), ("combined", 2048, 4, 128)])
def test_per_gpu_share_collect_and_replicated_fused_update(two_ranks, task, n_envs, n_steps, batch):
    a, b = two_ranks("sharded_training", task=task, n_envs=n_envs, n_steps=n_steps, batch_size=batch, n_epochs=2, iterations=3)
    for r in (a, b):
        f = r["facts"]
        assert f["fused_collect"] and f["graphs"] and f["replicated"] and f["fused_update"] and f["use_fused_norm"], f
        assert f["stats_sync"] == "rollout" and f["graph_captured"]
        assert r["checks"] == [0.0, 0.0, 0.0], "replicas diverged"
        assert r["finite"] and r["num_timesteps"] == 3 * n_steps * n_envs * 2
        assert r["allgather_bytes"] == n_steps * n_envs * (28 + 7) * 4
        assert r["counters"]["launches"] == 3 * n_steps
    assert np.array_equal(a["weights"], b["weights"]) and np.array_equal(a["stats"], b["stats"])
    assert a["stats"][2 * 28] == pytest.approx(1e-4 + (3 * n_steps + 1) * n_envs * 2)          # observations of BOTH ranks counted
    assert not np.array_equal(a["first_obs"], b["first_obs"]), "the ranks must simulate different env shards"


def test_allreduce_mode_keeps_the_replicas_identical_on_the_gpu(two_ranks):
    """with PPOConfig.dist_update = "allreduce" every rank trains on its LOCAL shard and the gradients are
    all-reduced per minibatch -- that exchange only exists on the torch path, so the fused fw_ppo_update (which has none) must
    not be taken there even though its shape conditions hold.  Replicas stay bit-identical."""
    a, b = two_ranks("sharded_training", task="waypoints", n_envs=512, n_steps=4, batch_size=128, n_epochs=2, iterations=2,
                     dist_update="allreduce")
    for r in (a, b):
        assert not r["facts"]["replicated"] and not r["facts"]["fused_update"], r["facts"]
        assert r["checks"] == [0.0, 0.0], "replicas diverged"
        assert r["finite"]
    assert np.array_equal(a["weights"], b["weights"])


def test_every_rank_raises_when_one_rank_reports_a_collector_timeout(two_ranks):
    """The status word of fw_collect_step is rank-local; the update is a collective point.  If only the rank with the non-zero word
    raised, the others would sit in the rollout all-gather: the ranks agree on the union of their words first."""
    a, b = two_ranks("status_agreement", n_envs=512, fallback=False)
    assert "another rank of the job reported status word 2" in a["msg"], a["msg"]
    assert "status word 2" in b["msg"] and "fold wave" in b["msg"], b["msg"]
    assert "gave up on another rank" in a["msg2"], a["msg2"]
    assert "fw_ppo_update gave up inside the launch" in b["msg2"], b["msg2"]
    assert a["checksum"] == 0.0 and b["checksum"] == 0.0      # rank 0 discarded its (completed) update with rank 1's: still replicas


def test_every_rank_falls_back_together_when_one_rank_reports_a_collector_timeout(two_ranks):
    """Default (PPOConfig.collect_fallback): the ranks agree on the union of their words, ALL of them take the void rollout back --
    statistics, timestep counter -- re-arm on the three-launch collector and collect the rollout again; nobody raises, nobody is left
    in a collective, the replicas stay bit-identical and count every sample once."""
    a, b = two_ranks("status_agreement", n_envs=512, fallback=True)
    for r in (a, b):
        assert r["msg"] == "no error", r["msg"]
        f = r["after"]
        assert f["fallbacks"] == 1 and not f["one_launch"] and len(f["warned"]) == 1, f
        assert f["num_timesteps"] == 4 * 512 * 2                   # ONE rollout of both ranks: the void one was taken back
        assert f["obs_count"] == pytest.approx(1e-4 + (4 + 1) * 512 * 2)
        assert f["checksum"] == 0.0 and r["checksum"] == 0.0
    assert "another rank of the job reported status word 2" in a["after"]["warned"][0]


def test_one_rank_job_over_rccl_runs_every_collective_on_device_tensors(launcher):
    """The other half of the rehearsal (the two-rank jobs above run over gloo and stage through the host): ONE rank over RCCL with
    FW_DIST_FORCE=1, so that the `backend == "nccl"` branches -- weight broadcast, statistics all-reduce, all_gather_into_tensor of the
    rollout shard, the gradient all-reduce of dist_update="allreduce", train()'s collective status look, the evaluation broadcast --
    run on the real library, beside the single-process job on the same seed."""
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    job = [sys.executable, os.path.join(root, "tests", "rccl_one_rank_job.py")]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    rdv = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FW_DIST_FORCE": "1"}
    rc, out, err = launcher(job, env=rdv, timeout=600)
    assert rc == 0, err[-3000:]
    forced = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    rc, out, err = launcher(job, timeout=600)
    assert rc == 0, err[-3000:]
    plain = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert forced["initialised"] and forced["backend"] == "nccl" and not plain["initialised"]
    for mode in ("replicated", "allreduce"):
        f, p = forced[mode], plain[mode]
        assert f["sharded"] and not p["sharded"] and f["finite"] and f["fallbacks"] == 0 and f["checksum"] == 0.0
        assert f["replicated"] == (mode == "replicated") and f["timesteps"] == p["timesteps"] == 2 * 8 * 256
        if mode == "replicated":
            assert f["allgather_bytes"] == 8 * 256 * 35 * 4 and f["one_launch"] and f["graphs"] and f["eval_mean_reward"] == 2.0
        # same samples counted, same scale of weights; not the same bits: a sharded job exchanges the normaliser's statistics BETWEEN
        # rollouts (stats_sync="rollout") where the single-process job updates them at every step, and dist_update="allreduce"
        # steps through torch + all-reduce where the single-process job takes the fused update
        assert f["count"] == p["count"] and f["weights_abs"] == pytest.approx(p["weights_abs"], rel=2e-2)
