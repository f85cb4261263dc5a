import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_tunables():
    """The library reads its NTR_* environment tunables once; tests that change one call nt.set_tunables()
    themselves, and every test starts from the environment as it is now."""
    try:
        import ntrace_amd as nt
        nt.lib().ntr_tunables_reload()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Make sure the oracle (gcc) and, when hipcc is present, the product library exist."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = os.path.join(ROOT, "ntrace_amd", "libntrace_amd.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ntrace_amd", "csrc")])
    yield
