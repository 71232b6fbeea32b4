import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def ctx():
    """One libhesaff_amd context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    import hesaff_amd
    c = hesaff_amd.HesaffContext(device=0)
    yield c
    c.close()
