"""tiddit_amd — MI355X-native (gfx950 HIP) implementation of TIDDIT's signal-aggregation and
clustering hot path behind the reference's own Python module API.

Modules mirror the reference package (``tiddit.tiddit_coverage`` -> ``tiddit_amd.tiddit_coverage``
...): same function names, arguments and return values; the arithmetic runs in hand-written HIP
kernels reached through the C ABI of ``libtiddit_hip.so`` (include/tiddit_hip.h).  There is no CPU
fallback: importing a compute module without the built library, or calling it without a GPU, raises.
"""
__version__ = "0.1.0"
