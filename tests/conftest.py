import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


# every ctx / workspace buffer the package hands to the library starts as 0xFF bytes (NaN in every float format): a kernel that
# reads what no kernel wrote fails a parity test instead of passing on whatever the allocator's block held before
os.environ.setdefault("SED_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
