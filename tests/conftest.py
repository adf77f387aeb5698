import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _torch_tensors_are_complete_before_the_library_sees_them(request):
    """GPU tests fill fresh torch tensors on torch's stream and hand them to calls on the library's own (non-blocking) streams: the Python binding then
    waits for torch's stream before every such call (capi.FENCE_TORCH).  Found by a test whose 0xAB fill landed AFTER vp_records_compact had written
    its slot headers once vp_frame had become faster; the C ABI itself makes no promise about other streams, as no HIP library does."""
    if request.node.get_closest_marker("gpu") is not None:
        import cases
        cases.vp.load_capi().FENCE_TORCH = True
    yield


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
