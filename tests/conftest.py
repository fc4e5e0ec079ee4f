import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library, built in-tree if stale (hipcc cross-compiles without a GPU)."""
    from cuttlefish_amd import build
    build.build()
    from cuttlefish_amd import api
    return api.load_library()


@pytest.fixture(scope="session")
def gpu_ctx(hip_lib):
    from cuttlefish_amd import api
    if api.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    ctx = api.Context(0)
    yield ctx
    ctx.close()
