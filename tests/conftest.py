import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (full golden file); run with RIMPHONY_SLOW=1")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    from rimphony_amd import _build
    _build.build_oracle()
    return oracle_bind.load("det")


@pytest.fixture(scope="session")
def oracle_libm():
    import oracle_bind
    from rimphony_amd import _build
    _build.build_oracle()
    return oracle_bind.load("libm")


@pytest.fixture(scope="session")
def gpu_ctx():
    from rimphony_amd import api
    ctx = api.Context(0)
    yield ctx
    ctx.close()
