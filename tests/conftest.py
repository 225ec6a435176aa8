import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (checker) and the product library exist."""
    import __graft_entry__ as g
    g.build_oracle()
    if not (os.path.exists(os.path.join(ROOT, "m17_sdr_amd", "libm17gpu.so"))
            and os.path.exists(os.path.join(ROOT, "tests", "compat", "libm17compat_harness.so"))):
        g.build()
    yield
