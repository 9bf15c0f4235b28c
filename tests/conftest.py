import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import c_oracle
    c_oracle.build_library()
    return c_oracle


@pytest.fixture(params=["group", "tpe"])
def step_kernel(request, monkeypatch):
    """Runs a GPU test once per step kernel (lane-group / thread-per-env; rg_create reads
    RG_STEP_KERNEL).  Configurations the thread-per-env kernel does not cover (N > 6) run the
    lane-group kernel in both instances."""
    monkeypatch.setenv("RG_STEP_KERNEL", request.param)
    return request.param


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The suites need the two in-tree libraries: librobogym_hip.so (hipcc cross-compiles without a
    GPU) and the C oracle.  Build whatever is missing or stale before the first test; the PRODUCT
    still fails loudly on its own when its library is absent (tests/test_host.py)."""
    from marbler_amd import build as hip_build
    if hip_build.needs_build():
        try:
            hip_build.hipcc_path()
        except RuntimeError as exc:   # no hipcc on this machine: the tests that need the library will say so
            print(f"[conftest] cannot build librobogym_hip.so: {exc}")
        else:
            hip_build.build()         # a compile error fails the session: never test a stale library
    from oracle import c_oracle
    c_oracle.build_library()
    yield
