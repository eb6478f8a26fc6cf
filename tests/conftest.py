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
# The suite drives every path of the library -- kernel variants, table geometries, fallbacks, injected failures -- through
# environment switches that only the TEST build of the library compiles in (krust_amd/csrc/ctx.hip.h, Knobs;
# `make -C krust_amd/csrc` builds both).  The product library (libkmerhip.so: what bench.py, smoke() and the kmerust binary
# load) has none of them; tests/test_gpu_product_lib.py holds it to the oracle as it ships.  Set before krust_amd is imported.
if os.path.exists(os.path.join(ROOT, "krust_amd", "lib", "libkmerhip_testing.so")):
    os.environ.setdefault("KMERHIP_LIB", "libkmerhip_testing.so")


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
