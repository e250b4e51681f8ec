import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# a variable left over from tools/ (diagnostic runs) must not change what the gates test
os.environ.pop("SPH_HIP_ARITH", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (oracle/liboracle.so) — test infrastructure, the checker."""
    from oracle.oracle import Oracle, build
    build(ref=None if os.path.isdir("/root/reference/src") else False)
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own compiled sph.cpp (oracle/_ref/libsphref.so), when loadable."""
    from oracle.oracle import Reference, reference_available
    if not reference_available():
        pytest.skip("oracle/_ref/libsphref.so not built/loadable on this machine")
    return Reference()


@pytest.fixture(scope="session")
def hiplib():
    """libsph_hip.so — the product. Built on demand; never falls back to anything."""
    from smoothed_particle_hydrodynamics_amd import build_library, load_library
    build_library()
    return load_library()
