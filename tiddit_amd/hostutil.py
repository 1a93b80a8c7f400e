"""Host-side helpers shared by the drop-in modules."""
import contextlib
import gc


@contextlib.contextmanager
def quiet_gc(freeze=False):
    """The row tables of the signal / cluster stages are millions of small, acyclic lists and dicts; CPython's cyclic collector walks
    all of them again and again while they are built (a third of the host time of `tiddit --sv` on a 48 M-read BAM).  Reference
    counting still frees everything; the collector is switched back on afterwards if it was on.

    freeze=True (the command line, which owns its process): ``gc.freeze()`` first — what the job left alive moves to the permanent
    generation, or the first allocation after ``gc.enable()`` starts a full collection over all of it (0.25 s at 240 M reads, 0.8 s
    at 600 M).  Frozen objects are still freed by their reference counts; the next ``quiet_gc`` (or :func:`thaw`) hands them back to
    the collector.  The library entry points (``tiddit_signal.main`` ...) do not freeze: a host's own objects are none of their business."""
    global _FROZEN
    was = gc.isenabled()
    gc.disable()
    if _FROZEN:
        gc.unfreeze()
        _FROZEN = False
    try:
        yield
    finally:
        if was:
            if freeze:
                gc.freeze()
                _FROZEN = True
            gc.enable()


_FROZEN = False


def thaw():
    """undo the ``gc.freeze()`` of ``quiet_gc(freeze=True)``"""
    global _FROZEN
    gc.unfreeze()
    _FROZEN = False


class PinnedPool:
    """Grow-only pinned host buffers (``tdt_host_alloc``) handed out as numpy arrays: columns built in them cross PCIe by DMA
    without a staging copy.  One pool per purpose; ``take(name, n, dtype)`` returns a view of at least n elements that stays valid
    until the next ``take`` of the same name with a larger size (or ``close``)."""

    def __init__(self):
        self._buf = {}

    def take(self, name, n, dtype):
        import ctypes
        import numpy
        from . import _native
        dtype = numpy.dtype(dtype)
        need = max(int(n), 1) * dtype.itemsize
        ent = self._buf.get(name)
        if ent is None or ent[1] < need:
            lib = _native.load()
            if ent is not None:
                lib.tdt_host_free(ctypes.c_void_p(ent[0]))
            cap = need + need // 4 + 4096
            p = ctypes.c_void_p()
            _native.check(lib.tdt_host_alloc(cap, ctypes.byref(p)))
            ent = (p.value, cap)
            self._buf[name] = ent
        raw = (ctypes.c_char * ent[1]).from_address(ent[0])
        return numpy.frombuffer(raw, dtype=dtype, count=int(n))

    def close(self):
        import ctypes
        from . import _native
        lib = _native.load()
        for p, _ in self._buf.values():
            lib.tdt_host_free(ctypes.c_void_p(p))
        self._buf.clear()
