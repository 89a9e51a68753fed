import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def S():
    """The product package (loads libsabc_hip.so; building it needs hipcc only)."""
    import sabc_amd
    sabc_amd.build()
    return sabc_amd


@pytest.fixture(scope="session")
def gpu(S):
    if S.lib().sabc_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    return 0


SEED = 20241220


def y_obs_mean():
    return float(np.random.default_rng(SEED).normal(1.5, 1.0, 100).mean())
