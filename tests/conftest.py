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
    from tests import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def gpu_ctx():
    from raytracing_amd import capi
    ctx = capi.Context(0)  # raises loudly if the library or a gfx950 device is missing — no fallback
    yield ctx
    ctx.close()
