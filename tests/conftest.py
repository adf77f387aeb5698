import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    import oracle as O
    O.build(ref=os.path.isdir("/root/reference/kernel"))
    return O.Port(threads=4)


@pytest.fixture(scope="session")
def ref():
    import oracle as O
    if not O.Ref.available():
        pytest.skip("oracle/_ref/libvpref.so not built (needs /root/reference)")
    return O.Ref()
