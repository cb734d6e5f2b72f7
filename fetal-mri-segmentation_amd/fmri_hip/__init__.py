"""ctypes binding of libfmri_hip.so (the C ABI declared in include/fmri_hip.h) + the U-Net execution engine.

The product path has NO CPU fallback: `lib()` raises if the shared library is missing, and every op raises on a
non-zero return code.
"""
from ._lib import lib, LibraryMissing, FmriError, F32, BF16, ACT_NONE, ACT_RELU, ACT_LEAKY, IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA  # noqa: F401
