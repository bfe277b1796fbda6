"""deciphon_amd -- MI355X-native Viterbi scan path of Deciphon.

The package is a thin ctypes binding over `lib/libdeciphon_hip.so` (HIP kernels for
gfx950 + the C ABI of `include/deciphon_hip.h`).  There is no CPU implementation
behind it: importing works anywhere, computing needs the built library and a GPU.
"""
from .hip import Engine, HipError, Window, device_count, encode, error_string, library_path, load_library, xtrans

__all__ = ["Engine", "HipError", "Window", "device_count", "encode", "error_string", "library_path",
           "load_library", "xtrans"]
