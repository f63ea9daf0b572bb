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
    """The CPU restatement (oracle/), used only as the checker."""
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def vo():
    """One HIP context for the whole GPU session (GPU tests run in one process)."""
    from visual_odometry_amd import api

    ctx = api.VoContext(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def frames():
    """Three consecutive synthetic KITTI-shaped frames (1241x376) + poses + K."""
    from visual_odometry_amd import synth

    return synth.make_sequence(0, 3)
