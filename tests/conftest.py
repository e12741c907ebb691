import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:          # (test files share problem builders: from test_engine_gpu import _problem)
    sys.path.insert(1, os.path.join(ROOT, "tests"))


_BLAS_LIMIT = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The float64 oracle is thousands of small NumPy / torch-CPU operations per step: on the GPU box's 256 hardware threads a BLAS
    # pool sized for the host makes every one of them slower (the bench's cpu_baseline calibrates the same way: 16 threads beat 64
    # and 256).  Cap the pools for the test process; the product sets no such limit.
    global _BLAS_LIMIT
    if (os.cpu_count() or 1) > 32:
        try:
            from threadpoolctl import threadpool_limits
            _BLAS_LIMIT = threadpool_limits(limits=16)
        except ImportError:
            pass
        try:
            import torch
            torch.set_num_threads(16)
        except ImportError:
            pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "packers_decode.npz"))


@pytest.fixture(scope="session")
def default_settings():
    import midi_vae_amd  # noqa: F401
    from midi_vae_amd.config import build_settings
    return build_settings()
