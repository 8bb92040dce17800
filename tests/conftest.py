import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def gemm_arith(request):
    """Run a test under a given GEMM arithmetic ('split' | 'f32') IN THIS PROCESS: the library's per-call switch
    (dsc_set_gemm_arithmetic), restored afterwards.  Use with
    ``@pytest.mark.parametrize("gemm_arith", ["split", "f32"], indirect=True)`` -- one pytest session then holds the goldens under both."""
    from diffuscene_amd import _lib
    prev = _lib.set_gemm_arithmetic(request.param)
    yield request.param
    _lib.set_gemm_arithmetic(prev)
