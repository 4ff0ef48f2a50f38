import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so); libpvlm.so links the system one with the same SONAME.
# Whichever is loaded first serves both: torch first works for both, libpvlm first leaves torch with "No HIP GPUs are
# available".  The tests that use torch next to the library (stream interop, 2-rank runs) therefore need torch loaded before
# any test module pulls in libpvlm.so — whatever subset of the suite is selected.
try:
    import torch  # noqa: F401
except Exception:  # the suite's CPU half does not need it
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/liboracle.so through ctypes."""
    from oracle import oracle as orc
    orc.build()
    return orc
