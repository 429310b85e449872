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


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the `gpu` tests instead of failing them
    (an explicit `-m gpu` still runs -- and fails loudly -- so a broken GPU box is noticed)."""
    if _has_gpu() or "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: needs a real MI355X (run with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
