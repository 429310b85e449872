import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def pli():
    """A Hip pipeline on device 0; GPU tests fail loudly if the extension is missing."""
    import lightmotif_amd as lm
    return lm.Pipeline.hip(0)


@pytest.fixture(scope="session")
def oracle():
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle
