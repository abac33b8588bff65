import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_det():
    from oracle_binding import Oracle, build
    build()
    return Oracle("det")


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle_binding import Oracle, build
    build()
    return Oracle("libm")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One HIP context for the whole GPU session (one process, one device)."""
    from atm_raytracer_amd import generators
    ctx = generators.Context(0)
    yield ctx
    ctx.close()
