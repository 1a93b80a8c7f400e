"""Host-side helpers shared by the drop-in modules."""
import contextlib
import gc


@contextlib.contextmanager
def quiet_gc():
    """The row tables of the signal / cluster stages are millions of small, acyclic lists and dicts; CPython's cyclic collector walks
    all of them again and again while they are built (a third of the host time of `tiddit --sv` on a 48 M-read BAM).  Reference
    counting still frees everything; the collector is switched back on afterwards if it was on."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
