"""MI355X-native blob-detection hot path of TIGERs-Mannheim/vision-processor.

Python here is plumbing for tests and the benchmark: `capi` binds the C ABI of libvp_hip.so (include/vp_hip.h),
`hostlib` binds the GPU-free C++ host side (synthetic frames, geometry producer), `configs` holds the standard
parameter sets.  The product is the HIP library and the C++ facade under host/.
"""
from . import configs, feeds, hostlib  # noqa: F401


def load_capi():
    """Lazy import: loading libvp_hip.so requires the HIP runtime."""
    from . import capi as _capi
    return _capi
