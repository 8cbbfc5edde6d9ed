import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# In the tests "accel" means what it says: the library's shortcut for small mixed-kind scenes (option "flat_below": a request for the tree is
# answered with the flat scan) is switched off, so the tree kernels of those scenes stay under test; the shortcut has its own test.
os.environ.setdefault("RTMI_FLAT_BELOW", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle_f32():
    from oracle.oracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def cover_small():
    """cover scene n=3 (a few dozen spheres) -- small enough for the CPU oracle in seconds"""
    import raytrace_clj_amd as r
    return r.scene.make_random_scene(200, 100, 3, False)


@pytest.fixture(scope="session")
def cover11():
    import raytrace_clj_amd as r
    return r.scene.make_random_scene(200, 100, 11, False)


@pytest.fixture(scope="session")
def cover11_moving():
    import raytrace_clj_amd as r
    return r.scene.make_random_scene(200, 100, 11, True)
