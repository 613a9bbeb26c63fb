import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes ``Library`` class as the product (prefix ``oracle_``).
    Test infrastructure: only tests/, smoke() and bench.py's cpu_baseline leg may load it."""
    from oracle_binding import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library (csrc/libtrajopt_hip.so).  Must exist; must see a GPU for -m gpu tests."""
    # torch first: it ships its own HIP runtime; whichever libamdhip64 is loaded first serves the whole process, and a torch
    # initialised AFTER the system runtime finds no devices (the RCCL gather test needs torch device tensors)
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    import trajopt_amd as T
    lib = T.load_hip_library()
    return lib
