import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native libraries are git-ignored build products: build them if a fresh checkout lacks them
    # (hipcc cross-compiles gfx950 without a GPU).  Up-to-date libraries are left alone.
    from ray_marching_amd import build
    if not (os.path.exists(build.HIP_SO) and os.path.exists(build.HOST_SO)):
        build.build_all()


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure, never part of the product path)."""
    from oracle import cbind
    cbind.build()
    cbind.lib()
    return cbind
