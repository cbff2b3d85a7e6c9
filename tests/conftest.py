import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the oracle is test infrastructure: only tests (and smoke/bench's baseline leg) import it
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box bring PyTorch's HIP runtime up FIRST.  The engine library links the system ROCm
    runtime, PyTorch ships its own copy; when the engine's runtime opens the device first, PyTorch's
    later initialisation in the same process reports "No HIP GPUs are available" (seen on the MI355X
    boxes; the other order works, and bench.py / smoke() use it too)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def have_ref(name):
    p = os.path.join(REF_DIR, name)
    return os.path.isfile(p) and os.access(p, os.X_OK)
