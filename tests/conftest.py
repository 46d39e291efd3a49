"""pytest configuration: the ``gpu`` marker and import paths.

``-m "not gpu"``: oracle vs golden fixtures, host logic, C-ABI symbol table (no GPU needed).
``-m gpu``: parity tests proper -- the HIP path through the C-ABI against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aind-exaspim-image-compression_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (built on demand with gcc)."""
    from oracle import bm4d_oracle
    bm4d_oracle.build()
    return bm4d_oracle


@pytest.fixture(scope="session")
def ctx():
    """The process's exabm4d context on cuda:0; fails loudly without the .so or a GPU."""
    from aind_exaspim_image_compression import _native
    return _native.context(0)
