import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pyflyt_drone_amd  # noqa: E402,F401  (import shim registers the package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import fw_oracle
    fw_oracle.build()
    return fw_oracle


@pytest.fixture(scope="session")
def K():
    from pyflyt_drone_amd import config
    return config
