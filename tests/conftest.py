import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cornerstone-octree_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    if not os.path.exists(orc.Oracle.libpath):
        orc.build("liboracle")
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    """the reference's own CPU code (oracle/_ref); only present where /root/reference was available at build time"""
    from oracle import oracle as orc

    if not orc.reference_available():
        if os.path.isdir("/root/reference/include/cstone"):
            orc.build("ref")
        else:
            pytest.skip("oracle/_ref not built (no /root/reference here)")
    return orc.Reference()


@pytest.fixture(scope="session")
def hip():
    """the product: C-ABI library loaded through ctypes (fails loudly if it is not built)"""
    import cstone_amd

    return cstone_amd.load()
