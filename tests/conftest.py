import os
import sys

import pytest

# torch ships its own copy of the HIP runtime (same soname as the system one libnos_hip.so links to): whichever is
# loaded first serves the whole process.  Load torch first so that tests which hand torch tensors / streams to the C ABI
# share one runtime with it, independent of test selection order.
try:  # pragma: no cover - environment dependent
    import torch  # noqa: F401
except Exception:  # torch is plumbing only; the suite runs without it
    torch = None

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
