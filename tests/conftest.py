import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.set_threads(int(os.environ.get("TSLO_THREADS", "4")))
    return pyoracle


@pytest.fixture(autouse=True)
def _collect_dead_scenes(request):
    """GPU tests: scene objects of earlier tests (reference cycles between a scene, its bodies and its gripper) are collected before the next
    test starts, so that their engine contexts -- and with them the device's dataflow token (csrc/direct_host.hpp) -- are released."""
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
    yield
