"""pytest configuration: markers and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU).
`-m gpu`       : parity tests proper; call the HIP path through the C ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is the checker; build it if the .so is not there yet."""
    so = os.path.join(ROOT, "oracle", "libkmer_oracle.so")
    src = os.path.join(ROOT, "oracle", "kmer_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return so


@pytest.fixture(scope="session")
def fixtures_dir():
    return os.path.join(ROOT, "tests", "fixtures")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
