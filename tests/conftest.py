import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def lib():
    """libbliss_amd.so, built in-tree if missing.  Never falls back to anything else."""
    import bliss_amd
    if not os.path.exists(bliss_amd._lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "bliss_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)
    return bliss_amd.load()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    assert lib.bl_amd_init(0) == 0
    return lib
