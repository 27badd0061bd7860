import os
import sys
import pytest

# PyTorch-ROCm ships its own copy of the HIP runtime; tests that use torch tensors next to libfasts_hip.so need
# torch's copy loaded first (same soname: the library then binds to it), or torch finds no GPU afterwards.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: test needs the compiled reference (oracle/_ref), skipped when absent")
