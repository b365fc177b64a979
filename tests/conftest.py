import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests bind the built libraries; in a fresh checkout (the .so files are not in git) build them first -- hipcc cross-compiles
    gfx950 without a GPU, the oracle is plain g++."""
    if not (os.path.exists(os.path.join(ROOT, "vk_raytrace_amd", "libptmi.so")) and os.path.exists(os.path.join(ROOT, "oracle", "liborc.so"))):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc_lib():
    from tests import orc
    return orc.lib()
