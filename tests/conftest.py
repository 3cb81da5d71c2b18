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
    """CPU oracle (test infrastructure): oracle/libhjoracle.so via ctypes."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hj():
    """One hjgpu context for the whole session.  No fallback: if the HIP library
    or the GPU is missing this raises and every gpu test errors out loudly."""
    # torch bundles its own HIP runtime: when both live in one process torch has to
    # initialise first (bench.py does the same), otherwise it finds no device
    try:
        import torch
        torch.cuda.init()
    except ImportError:
        pass
    import hash_join_codes_knl_amd as H
    ctx = H.HjGpu()
    info = ctx.device_info()
    assert "gfx950" in info["arch"], "these kernels are built for gfx950 only, found %r" % info
    yield ctx
    ctx.close()
