import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def hip_lib():
    """The real HIP library (built by __graft_entry__.build()). No fallback: missing library == failure."""
    from qm_door_amd import abi
    return abi.load_library()


@pytest.fixture(scope="session")
def interface(hip_lib):
    from qm_door_amd import api
    return api.QMInterface(lib=hip_lib)


@pytest.fixture(scope="session")
def oracle(interface):
    import support
    return support.Oracle(interface.problem)
