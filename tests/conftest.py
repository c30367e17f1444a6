import os
import sys

import pytest
import torch  # noqa: F401  -- before the engine library: torch ships its own HIP runtime under the same soname as
#                 the system's, and whichever is loaded first serves both; torch only finds the GPU through its own

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import gsw_oracle

    gsw_oracle.build()
    return gsw_oracle
