import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pyflyt_drone_amd  # noqa: E402,F401  (import shim registers the package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# ---- two helper processes for the multi-rank GPU tests, started before this process touches the GPU (see tests/mp_jobs.py)
_MP = {}


def pytest_sessionstart(session):
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mp_jobs
    ctx = mp.get_context("spawn")
    _MP["res"] = ctx.Queue()
    _MP["jobs"] = [ctx.Queue() for _ in range(2)]
    _MP["procs"] = [ctx.Process(target=mp_jobs.serve, args=(r, 2, _MP["jobs"][r], _MP["res"]), daemon=True) for r in range(2)]
    _MP["launch_q"], _MP["launch_res"] = ctx.Queue(), ctx.Queue()
    _MP["procs"].append(ctx.Process(target=mp_jobs.launch_serve, args=(_MP["launch_q"], _MP["launch_res"]), daemon=True))
    for p in _MP["procs"]:
        p.start()


def pytest_sessionfinish(session, exitstatus):
    for q in _MP.get("jobs", []) + ([_MP["launch_q"]] if "launch_q" in _MP else []):
        q.put(None)
    for p in _MP.get("procs", []):
        p.join(timeout=20)


@pytest.fixture
def two_ranks():
    """run(name, **kwargs) -> [result of rank 0, result of rank 1] of tests/mp_jobs.<name> as a world_size-2 gloo job."""
    if not _MP:
        pytest.skip("helper ranks are only started for -m gpu sessions")
    import socket

    def run(name, timeout=600, **kwargs):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        for q in _MP["jobs"]:
            q.put((name, port, kwargs))
        res = sorted([_MP["res"].get(timeout=timeout) for _ in range(2)], key=lambda r: r[0])
        for rank, status, out in res:
            assert status == "ok", f"rank {rank} failed:\n{out}"
        return [out for _, _, out in res]
    return run


@pytest.fixture
def launcher():
    """run(argv, env={}, timeout=600) -> (returncode, stdout, stderr) of a program started by a helper process that never
    touches the GPU (the pytest process has initialised it and must not start programs itself on the GPU boxes)."""
    if "launch_q" not in _MP:
        pytest.skip("the launcher helper is only started for -m gpu sessions")

    def run(argv, env=None, timeout=600):
        _MP["launch_q"].put((list(argv), dict(env or {}), timeout))
        status, out = _MP["launch_res"].get(timeout=timeout + 60)
        assert status == "ok", out
        return out
    return run


@pytest.fixture(scope="session")
def oracle():
    from oracle import fw_oracle
    fw_oracle.build()
    return fw_oracle


@pytest.fixture(scope="session")
def K():
    from pyflyt_drone_amd import config
    return config
