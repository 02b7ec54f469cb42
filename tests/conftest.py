import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The ring kernel's 3x3 mode is dispatched from 2^18 output pixels up (where it wins); the tests lower that threshold so that
# moderately sized cases and the model-level tests exercise it too.  Read once by libocrvi, at the first 3x3 launch.
os.environ.setdefault("OCRVI_RING_CONV3_MIN_M", "16384")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
