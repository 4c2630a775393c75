import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; built with gcc on first use)."""
    from oracle import loader
    loader.build()
    return loader


@pytest.fixture(scope="session")
def ctx():
    """One single-device context for the whole GPU session (cuda:0)."""
    from nonlinear_optimizer_for_slam_amd import Context
    c = Context((0,))
    yield c
    c.close()
