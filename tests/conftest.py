import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One tdc_gpu context for the whole GPU session.  No skip-on-missing-GPU: on a GPU box a missing library or
    device is a failure, not a skip."""
    import tudocomp_amd as T
    ctx = T.Context(0)
    yield ctx
    ctx.close()
