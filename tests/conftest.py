import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EARTH = os.path.join(GOLDEN, "earth_synth.ppm")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """Make sure the shared libraries exist (hipcc cross-compiles gfx950 without a GPU)."""
    import raytracingoneweekendapplication_amd as rt
    from oracle import orc

    if not (os.path.exists(rt.HIP_LIB_PATH) and os.path.exists(rt.HOST_LIB_PATH) and os.path.exists(orc.LIB_PATH)):
        import __graft_entry__

        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def rt():
    import raytracingoneweekendapplication_amd as rt

    return rt


@pytest.fixture(scope="session")
def orc():
    from oracle import orc

    return orc


@pytest.fixture(scope="session")
def renderer(rt):
    """A Renderer on device 0 (GPU tests only; raises loudly when no gfx950 device exists)."""
    return rt.Renderer(0)
