"""roctx ranges around the stages of the pipeline (tracing): `rocprofv3 --marker-trace` shows them next to the
kernels.  Bound lazily with ctypes; a no-op when no roctx library is present (nothing else depends on it)."""
import contextlib
import ctypes

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = False
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except (OSError, AttributeError):
                continue
    return _lib


@contextlib.contextmanager
def stage(name):
    lib = _load()
    if lib:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib:
            lib.roctxRangePop()
