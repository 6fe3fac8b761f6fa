"""bench.py end to end on the GPU box: the one JSON line the driver parses, at N = 1 and -- as a rehearsal of the multi-GPU
launch on the one GPU there is -- `python bench.py --gpus 2` started bare, which launches its own two ranks (both on cuda:0,
barrier / max-reduce over gloo because RCCL refuses two ranks on one device)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return lines[0]


def test_bench_line_n1_carries_the_contract_fields(launcher):
    rc, out, err = launcher([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "30", "--no-cpu-baseline"])
    assert rc == 0, err
    b = _line(out)
    assert b["n_gpus"] == 1 and b["steps"] == 20 and b["warmup"] == 5 and b["repeats"] == 30
    assert b["unit"] == "env-steps/s" and b["dtype"] == "f64" and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["ms_per_step_min"] <= b["ms_per_step"] <= b["ms_per_step_max"]
    assert b["value"] == pytest.approx(4096 * 1e3 / b["ms_per_step"], rel=1e-9)
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["algorithmic_bytes_per_launch"] == 94 * 8 * 4096
    assert b["value"] > 1e6                                  # BASELINE.json's target


def test_bare_gpus_2_launches_its_own_ranks_on_the_gpu(launcher):
    rc, out, err = launcher([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "10"],
                            env={"FW_BENCH_BACKEND": "gloo", "FW_BENCH_SINGLE_DEVICE": "1"})
    assert rc == 0, err
    b = _line(out)
    assert b["n_gpus"] == 2 and "cpu_baseline" not in b and b["config"]["parallelism"] == "env-shard x2"
    assert b["value"] == pytest.approx(2 * 4096 * 1e3 / b["ms_per_step"], rel=1e-9)
    assert "update_allgather" in b and b["update_allgather"].get("samples_per_rank") == 16 * 4096
    # the world > 1 form of the line: every rank's own clock, and the sharded collector object with the
    # reference's samples per update held (n_steps = 65 536 / (4096 x world))
    assert len(b["per_rank_kernel_ms_per_step"]) == 2 and b["per_rank_kernel_ms_per_step_min"] <= b["per_rank_kernel_ms_per_step_max"]
    c = b["collector"]
    assert "error" not in c, c
    assert c["world"] == 2 and c["n_steps"] == 8 and c["samples_per_update"] == 65536 and c["replica_checksum"] == 0.0
    assert c["collect_fallbacks"] == 0 and c["end_to_end_env_steps_per_s"] > 1e5 and c["update_allgather_bytes_per_rank"] == 8 * 4096 * 35 * 4


def test_one_rank_job_over_rccl_prints_the_multi_gpu_form_of_the_line(launcher):
    """FW_DIST_FORCE=1: one rank, but through the process group -- the RCCL branches of bench.py (rank count by all-reduce on the GPU,
    per-rank gather, all_gather_into_tensor of the update shard, the sharded collector object) on the real library, which the
    two-rank gloo rehearsal above cannot reach."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    rc, out, err = launcher([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "5", "--no-cpu-baseline"],
                            env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FW_DIST_FORCE": "1"})
    assert rc == 0, err[-3000:]
    b = _line(out)
    assert b["n_gpus"] == 1 and b["ranks_seen"] == 1 and b["backend"].startswith("RCCL") and b["value"] > 1e6
    g = b["update_allgather"]
    assert "error" not in g, g
    assert g["collective"].startswith("all_gather_into_tensor (RCCL)") and g["bytes_per_rank"] == 16 * 4096 * 35 * 4
    assert len(b["per_rank_kernel_ms_per_step"]) == 1
    c = b["collector"]
    assert "error" not in c, c
    assert c["world"] == 1 and c["n_steps"] == 16 and c["samples_per_update"] == 65536 and c["replica_checksum"] == 0.0 and c["collect_fallbacks"] == 0
    assert c["one_launch_collect"] and c["end_to_end_env_steps_per_s"] > 5e5


def test_gpus_mismatch_is_an_error_not_a_one_gpu_line(launcher):
    rc, out, _ = launcher([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env={"WORLD_SIZE": "1", "RANK": "0"})
    b = _line(out)
    assert rc != 0 and "error" in b and "n_gpus" not in b
