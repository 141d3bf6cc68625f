import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product)."""
    from oracle import fv_oracle

    return fv_oracle


@pytest.fixture(scope="session")
def fv():
    """The product package (finitevolume.jl_amd/), loaded as `fvamd`; needs libfvhip.so."""
    from __graft_entry__ import load_package

    return load_package()
