import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _libavc_is_built():
    """Build libavc.so in-tree when it is missing or older than its sources (hipcc cross-compiles without a GPU; a no-op when
    the library that travelled with the snapshot is current).  This only makes sure the ARTEFACT exists: the product path
    still has no fallback and raises when it cannot load the library."""
    try:
        from avatarclip_amd import build
        build.build()
    except Exception as e:   # no hipcc on this machine: the tests that need the library will say so themselves
        print("libavc build skipped:", e)
    yield
