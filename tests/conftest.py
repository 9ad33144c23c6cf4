import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(autouse=True)
def _seed_global_rng(request):
    """Unseeded calls (RaySource.create_rays, the remainder of the per-source ray split, focus-search samples) draw their
    seeds from NumPy's global generator like the reference does: pin it per test, so that the statistical checks see
    the same rays on every run."""
    import zlib

    import numpy as np
    np.random.seed(zlib.crc32(request.node.nodeid.encode()) & 0x7fffffff)
    yield
