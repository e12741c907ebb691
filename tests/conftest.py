import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:          # (test files share problem builders: from test_engine_gpu import _problem)
    sys.path.insert(1, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "packers_decode.npz"))


@pytest.fixture(scope="session")
def default_settings():
    import midi_vae_amd  # noqa: F401
    from midi_vae_amd.config import build_settings
    return build_settings()
