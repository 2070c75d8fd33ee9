import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_api
    return oracle_api.load()


@pytest.fixture(scope="session")
def gpu_ctx_ok():
    """Fails (does not skip) when the HIP extension is missing or no device opens: GPU tests
    must never pass on a fallback."""
    from ti_raytrace_amd import _native
    _native.lib()
    assert _native.device_count() >= 1
    return True
