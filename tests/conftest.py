import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "covariancefunctions.jl_amd"))   # product: `import covgram`
sys.path.insert(0, os.path.join(ROOT, "oracle"))                       # checker: `import covgram_oracle` (tests only)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def cg():
    import covgram
    return covgram


@pytest.fixture(scope="session")
def oracle():
    import covgram_oracle
    return covgram_oracle
