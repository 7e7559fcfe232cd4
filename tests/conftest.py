import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
HERE = os.path.join(ROOT, "tests")
if HERE not in sys.path:
    sys.path.insert(0, HERE)                 # tests/parity.py (comparison helpers)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Largest relative errors the parity helper saw, per quantity (tests/parity.py: MEASURED)."""
    try:
        import parity
    except ImportError:
        return
    if parity.MEASURED:
        terminalreporter.write_line("parity: largest relative error per quantity (close_rel) -- "
                                    + ", ".join(f"{k}: {v:.2e}" for k, v in sorted(parity.MEASURED.items())))
