import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with g++."""
    from oracle import orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def emu_lib():
    """CPU harness that runs the kernel source (srt_core.h) one lane at a time -- tests only."""
    import helpers
    return helpers.build_emu()


@pytest.fixture(scope="session")
def gpu_renderer_factory():
    """Renderer factory bound to the real libsrt.so on cuda:0; fails loudly if the library or the GPU is missing."""
    import simple_raytracer_amd as srt

    def make():
        return srt.Renderer(device=0)
    return make
