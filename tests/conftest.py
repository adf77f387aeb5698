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
def _torch_tensors_are_ordered_before_the_library_sees_them(request):
    """GPU tests fill fresh torch tensors on torch's stream and hand them to calls on the library's own (non-blocking) streams: the Python binding then
    orders torch's stream (of the tensor's device) before every such call with a device-side event (capi.FENCE_TORCH -> vp_stream_wait_for; no host wait,
    so the suite still submits its calls back to back as bench.py and real callers do).  Found by a test whose 0xAB fill landed AFTER vp_records_compact
    had written its slot headers once vp_frame had become faster; the C ABI itself makes no promise about other streams, as no HIP library does."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import cases
    capi = cases.vp.load_capi()
    capi.FENCE_TORCH = True
    try:
        yield
    finally:
        capi.FENCE_TORCH = False
        del capi._pending[:]


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
