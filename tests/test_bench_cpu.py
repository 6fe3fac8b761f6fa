"""bench.py's launch contract, checked without a GPU: `--gpus N` is honoured whichever way the script
is started.  FW_BENCH_DRY=1 is the launcher-rehearsal switch: ranks rendezvous, barrier and max-reduce over gloo, nothing is
measured and the line says so (`"dry_run": true, "value": null`); the measuring path itself needs the HIP device and is covered
by the -m gpu tests."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bare_gpus_2_self_launches_two_ranks_and_rank_0_prints_one_line():
    rc, lines, err = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"], FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo", FW_BENCH_SINGLE_DEVICE="1")
    assert rc == 0, err
    assert len(lines) == 1, lines                         # ONE JSON line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["max_rank_plus_one"] == 2.0 and lines[0]["dry_run"] is True
    assert lines[0]["steps"] == 20 and lines[0]["warmup"] == 5
    # the ranks are counted over the collective backend itself (an all-reduce(SUM) of ones): what lets the driver verify that
    # N ranks really met -- over RCCL on its 8-GPU node
    assert lines[0]["ranks_seen"] == 2 and lines[0]["backend"] == "gloo"


def test_bare_gpus_8_self_launches_eight_ranks_the_drivers_scaling_form():
    """The N = 8 line of the driver's scaling run, rehearsed end to end without a GPU: eight rank processes, one rendezvous on
    127.0.0.1, every rank counted over the backend, the per-rank gather that carries each rank's own clock, ONE line."""
    rc, lines, err = _run(["--gpus", "8", "--steps", "50", "--warmup", "10"], FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo")
    assert rc == 0, err
    assert len(lines) == 1, lines
    l = lines[0]
    assert l["n_gpus"] == 8 and l["ranks_seen"] == 8 and l["max_rank_plus_one"] == 8.0 and l["dry_run"] is True
    assert l["per_rank"] == [float(r + 1) for r in range(8)]              # rank order, every rank present


def test_a_rank_that_dies_takes_the_eight_rank_job_down_non_zero_within_the_timeout():
    """A dead rank must end the job -- the others would otherwise wait in the rendezvous or a collective until their own
    timeouts, and the driver's clock around the run would record a hang instead of a failure."""
    import time
    t0 = time.perf_counter()
    rc, lines, err = _run(["--gpus", "8"], FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo", FW_BENCH_DRY_DIE_RANK="5")
    assert rc != 0 and not any("value" in l and "error" not in l for l in lines), (rc, lines)
    assert time.perf_counter() - t0 < 120


def test_one_rank_job_goes_through_the_process_group_when_forced():
    """FW_DIST_FORCE=1 + a one-rank rendezvous (the form in which the -m gpu tests run the RCCL branches on a one-GPU box): the rank
    joins a process group, is counted over it and gathers its own figure -- here over gloo and dry."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    rc, lines, err = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"], FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo", FW_DIST_FORCE="1",
                          WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["ranks_seen"] == 1 and lines[0]["backend"] == "gloo" and lines[0]["per_rank"] == [1.0]
    # ... and without the switch the same environment is a plain single-process run
    rc, lines, err = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"], FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo",
                          WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert rc == 0 and lines[0]["backend"].startswith("none"), (lines, err)


def test_gpus_that_disagrees_with_the_world_size_exits_non_zero():
    """`--gpus 8` under a launcher that started one rank must never print `"n_gpus": 1`."""
    rc, lines, _ = _run(["--gpus", "8"], WORLD_SIZE="1", RANK="0", FW_BENCH_DRY="1")
    assert rc != 0 and len(lines) == 1 and "error" in lines[0] and "n_gpus" not in lines[0]
    assert lines[0]["n_gpus_requested"] == 8 and lines[0]["world_size"] == 1


def test_more_gpus_than_devices_is_refused_before_anything_is_launched():
    rc, lines, _ = _run(["--gpus", "2"])                 # no GPU in this container: 0 devices visible
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices are visible here")
    assert rc != 0 and "error" in lines[0] and "n_gpus" not in lines[0]


def test_under_a_launcher_the_ranks_agree_with_gpus():
    """The driver's form: torch.distributed.run provides WORLD_SIZE / RANK; --gpus equals the world size."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ, FW_BENCH_DRY="1", FW_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1"],
                       env=e, capture_output=True, text=True, timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stderr
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 7 and lines[0]["ranks_seen"] == 2


def test_learner_accounting_matches_the_documented_figures():
    """tools/learner_accounting.py is the numerator of every learner roofline (bench.py `collector.roofline`,
    profiles/r04_learner_pmc.json): the documented figures follow from it."""
    sys.path.insert(0, ROOT)
    from tools import learner_accounting as A
    c = A.collect_step(4096, 28, 94)
    # every buffer ONCE (the HBM roofline's numerator) + what the act waves fetch again out of the L2, under its own name; the sum is
    # round 4's "19.3 MB", which priced L2 hits as HBM bytes (its counter traffic was 0.54 x that)
    assert c["bytes"] == c["unique_bytes"] == sum(c["items"].values()) == 5_285_092
    assert c["l2_served_bytes"] == sum(c["l2_served_items"].values()) == 14_016_284
    assert c["unique_bytes"] + c["l2_served_bytes"] == 19_301_376
    cc = A.collect_close(4096, 28, 16)
    assert cc["unique_bytes"] == 2_801_412 and cc["l2_served_bytes"] == 6_202_620
    rc = A.roofline_hbm(c["bytes"], 32.67, 10.3e6, c["l2_served_bytes"])
    assert abs(rc["frac"] - 0.0202) < 2e-4 and rc["traffic_over_algorithmic"] > 1.0 and rc["l2_inclusive_gbps"] > 4 * rc["achieved"] * 0.9
    assert c["items"]["env_step (fw_step's words per env-step x N)"] == 94 * 8 * 4096
    u = A.ppo_update(10240, 128, 28)
    assert A.ppo_split(128) == (16, 8) and A.ppo_split(64) == (16, 4) and A.ppo_split(256) == (32, 8) and A.ppo_split(4096) == (64, 8)
    assert A.ppo_split(192) == (16, 8) and A.ppo_split(96) == (16, 4) and A.ppo_split(32) == (16, 2) and A.ppo_split(16) == (16, 1)      # csrc/fwsim_ppo.hpp ppo_split
    assert A.ppo_split(128, 4) == (32, 4) and A.ppo_split(256, 4) == (64, 4)      # round 4's cuts (FWSIM_PPO_RS=0)
    assert u["workgroups"] == 16 and abs(u["mfma_peak_tflops"] - 16 * 157.3 / 256) < 1e-12
    # forward + backward MACs of both networks per sample: (28 x 64 + 64 x 64 + 64 x KO) + (2 x 64 x KO + 2 x 64 x 64 + 28 x 64)
    macs = sum((28 * 64 + 64 * 64 + 64 * ko) + (2 * 64 * ko + 2 * 64 * 64 + 28 * 64) for ko in (4, 1))
    assert u["flops_per_minibatch"] == 2 * 128 * macs == 8_372_224
    assert A.ppo_update(5120, 64, 56)["workgroups"] == 8
    r = A.roofline_mfma(u["mfma_flops"], 102_876.2, 8 * 157.3 / 256)
    assert abs(r["frac"] - 0.1695) < 5e-4                      # profiles/r04_learner_pmc.json: fw_ppo_update_kernel, waypoints (round 4: 8 workgroups)
    assert A.render(4096, 32)["bytes"] == 33_554_432
    p = A.ppo_pack(10240, 128, 28)
    assert p["items"]["packed rows written ((D rounded up to 4) + 8 floats)"] == 10240 * 128 * 36 * 4
