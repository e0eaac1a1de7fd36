import os
import sys

import pytest

# the oracle's OpenMP regions are tiny in the tests; 100+ spinning threads (GPU box) only slow them down
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/, test infrastructure only)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth():
    from iceberg_tracking_code_amd import synth as s
    return s


@pytest.fixture(scope="session")
def ctx():
    """A GPU context big enough for the small parity cases."""
    from iceberg_tracking_code_amd import Context
    c = Context(1024, 768, n_slots=3, max_pts=1 << 16)
    yield c
    c.close()
