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


class _ContextProxy:
    """The session's GPU context, created on first use.  A test that needs the device for a child process (the first
    context on a device owns it: cooperative tail, full grids) asks for it with `with gpu_ctx.released():`; the context
    is closed for the duration and the NEXT use -- in this or any later test -- opens a fresh one.  No test depends on
    another one's clean-up."""

    def __init__(self):
        self._ctx = None

    def _get(self):
        if self._ctx is None:
            from rimphony_amd import api
            self._ctx = api.Context(0)
        return self._ctx

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def released(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.close_now()
            try:
                yield
            finally:
                self.close_now()          # whatever the body opened through the proxy does not outlive it either
        return cm()

    def close_now(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


@pytest.fixture(scope="session")
def gpu_ctx():
    proxy = _ContextProxy()
    yield proxy
    proxy.close_now()
