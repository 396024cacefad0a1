import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_artefacts():
    """The built library / tools / oracle normally travel with the tree; build them if a checkout lacks them."""
    import subprocess
    need = [os.path.join(ROOT, "tamcmc-c-_amd", "libtamcmc_accel.so"), os.path.join(ROOT, "bin", "cpptamcmc_hip"),
            os.path.join(ROOT, "bin", "getmodel_hip")]
    if not all(os.path.exists(f) for f in need):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tamcmc-c-_amd", "csrc"), "-j4"], check=True)
    if not any(f.endswith(".so") for f in os.listdir(os.path.join(ROOT, "oracle"))):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def accel_mod():
    import tamcmc_amd
    return tamcmc_amd
