import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def hostsim_library(path):
    """The host engine linked against the CPU stand-in for the kernels, bound with the package's own ctypes table.  The
    product loader accepts the HIP build only; the acceptance of the simulator lives here, under tests/."""
    from pymes_amd import _lib

    class HostsimLibrary(_lib.Library):
        BACKEND = "hostsim"
    return HostsimLibrary(path)


@pytest.fixture(scope="session")
def hostsim_lib():
    """Host engine linked against the CPU stand-in for the kernels (tests/hostsim): checks HOST logic only."""
    from pymes_amd import _lib
    d = os.path.join(ROOT, "tests", "hostsim")
    san = os.environ.get("PYMES_HOSTSIM_LIBRARY")      # a sanitizer build (tests/hostsim/run_sanitized.sh)
    if san:
        return hostsim_library(san)
    subprocess.run(["make", "-s", "-C", d], check=True)
    return hostsim_library(os.path.join(d, "_build", "libpymes_hostsim.so"))


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library (HIP, gfx950).  No fallback: a missing library is an error."""
    from pymes_amd import _lib
    return _lib.default_library()
