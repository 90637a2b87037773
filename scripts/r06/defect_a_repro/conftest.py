"""conftest of the FROZEN reproducer of defect (a): tests/test_gpu_configs.py as it stood at the end of round 5 (commit 79fc16b), whose
36 tests in one process are the shortest known trigger of the host SIGSEGV under SGA_GRAPH_DROP=destroy (scripts/r06/s0*_defect_a_*.sh)."""
import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")   # see sga_amd/__init__.py; must precede HIP init

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))      # scripts/r06/defect_a_repro -> repo root
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: on a fresh checkout compile them first (hipcc cross-compiles
    gfx950 without a GPU).  A failed build fails the session loudly; there is no CPU fallback."""
    pkg = os.path.join(ROOT, "improving-inference-for-neural-image-compression_amd")
    if not all(os.path.exists(os.path.join(pkg, n)) for n in ("libsga_hip.so", "libsga_hip_lab.so", "librans.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def gpu_out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d
