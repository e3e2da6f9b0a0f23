"""instarevive_amd — MI355X-native (gfx950) one-step image-restoration path with InstaRevive's inference surface.

Host side in Python (PyTorch-ROCm for device memory and streams), compute in hand-written HIP kernels behind the
C ABI of include/instarevive_hip.h. Importing this package never falls back to a CPU implementation.
"""
from ._lib import Context, NativeLibraryError, load_library, LIB_PATH  # noqa: F401

__all__ = ["Context", "NativeLibraryError", "load_library", "LIB_PATH"]
