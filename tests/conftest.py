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


@pytest.fixture(scope="session")
def experiments_lib(gpu_ctx_ok):
    """Tests of the EXPERIMENTS (cost-optimal wide collapse: built, bit-identical, not faster -- docs/HISTORY.md) need the
    library built with -DTIRT_EXPERIMENTS (`make -C ti_raytrace_amd/csrc experiments`, then TIRT_LIB_PATH=.../libtirt_exp.so): the product
    library does not carry them, and these tests are skipped on it."""
    from ti_raytrace_amd import _native
    ctx = _native.Context(0)
    try:
        ctx.set_option("wide_collapse", 1); ctx.set_option("wide_collapse", 0)
    except _native.TirtError as exc:
        if "TIRT_EXPERIMENTS" in str(exc):
            pytest.skip("the loaded libtirt.so is the product build (no -DTIRT_EXPERIMENTS)")
        raise
    return True
