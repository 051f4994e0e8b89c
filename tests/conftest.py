import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is test infrastructure: make sure it is built (gcc, ~2 s)
    so = os.path.join(ROOT, "oracle", "librt_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), so], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def cornell_dir():
    return os.path.join(ROOT, "tests", "golden", "cornell")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
