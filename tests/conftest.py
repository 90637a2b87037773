import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # see sga_amd/__init__.py; must precede HIP init

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def gpu_out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d
