"""ctypes binding of libtiddit_hip.so (include/tiddit_hip.h).  No CPU fallback: a missing library
or a missing GPU raises."""
import ctypes
import os
import sys
import threading


_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TIDDIT_HIP_LIB") or os.path.join(_HERE, "libtiddit_hip.so")

TDT_OK = 0
ERRORS = {-1: "TDT_E_ARG", -2: "TDT_E_HIP", -3: "TDT_E_RANGE", -4: "TDT_E_INEXACT", -5: "TDT_E_NOMEM",
          -6: "TDT_E_UNSUPPORTED", -7: "TDT_E_KEY"}

_lib = None
_lock = threading.Lock()


class TdtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code


# every symbol include/tiddit_hip.h declares: name -> (restype, argtypes)
_i, _i64, _sz, _dbl, _P = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_double, ctypes.c_void_p
_PP = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "tdt_version": (_i, []),
    "tdt_last_error": (ctypes.c_char_p, []),
    "tdt_build_flags": (ctypes.c_char_p, []),
    "tdt_device_count": (_i, [ctypes.POINTER(_i)]),
    "tdt_ctx_create": (_i, [_i, _PP]),
    "tdt_ctx_destroy": (None, [_P]),
    "tdt_ctx_sync": (_i, [_P]),
    "tdt_ctx_stream": (_P, [_P]),
    "tdt_ctx_set_stream": (_i, [_P, _P]),
    "tdt_ctx_bind_thread": (_i, [_P]),
    "tdt_cov_create": (_i, [_P, _P, _i, _i, _PP]),
    "tdt_cov_destroy": (None, [_P]),
    "tdt_cov_nbins": (_i, [_P, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i)]),
    "tdt_cov_scale_bits": (_i, [_P]),
    "tdt_cov_reset": (_i, [_P]),
    "tdt_cov_push": (_i, [_P, _i, _P, _P, _P, _P, _sz, _i]),
    "tdt_cov_push_device": (_i, [_P, _i, _P, _P, _P, _P, _sz, _i]),
    "tdt_cov_push_device_multi": (_i, [_P, _i, _P, _P, _P, _P, _P, _P, _i]),
    "tdt_cov_pack_device": (_i, [_P, _P, _P, _P, _P, _sz, _P]),
    "tdt_cov_push_packed_device_multi": (_i, [_P, _i, _P, _P, _P, _P, _i]),
    "tdt_cov_pack_binned_device": (_i, [_P, _i, _P, _P, _P, _P, _sz, _P]),
    "tdt_cov_push_binned_device_multi": (_i, [_P, _i, _P, _P, _P, _P, _P, _i]),
    "tdt_cov_total_bins": (_i, [_P, ctypes.POINTER(_i64)]),
    "tdt_cov_offset": (_i, [_P, _i, ctypes.POINTER(_i64)]),
    "tdt_cov_finish_all_device": (_i, [_P, _P]),
    "tdt_cov_finish_all": (_i, [_P, _P]),
    "tdt_cov_finish": (_i, [_P, _i, _P]),
    "tdt_cov_finish_device": (_i, [_P, _i, _P]),
    "tdt_cov_kept": (_i, [_P, ctypes.POINTER(_i64)]),
    "tdt_gc_bins": (_i, [_P, _P, _i64, _i, _dbl, _P]),
    "tdt_gc_bins_fasta": (_i, [_P, _P, _i64, _i64, _i, _i, _i, _dbl, _P]),
    "tdt_gc_bins_fasta_device": (_i, [_P, _P, _i64, _i64, _i, _i, _i, _dbl, _P]),
    "tdt_gc_bins_device": (_i, [_P, _P, _i64, _i, _dbl, _P]),
    "tdt_gc_bins_fasta_many": (_i, [_P, _P, _i64, _i, _P, _P, _P, _P, _P, _i, _dbl, _P, _P, _i64]),
    "tdt_dbscan": (_i, [_P, _P, _sz, _sz, _dbl, _i, _i, _P, ctypes.POINTER(_i64)]),
    "tdt_dbscan_y": (_i, [_P, _P, _sz, _sz, _dbl, _i, _i64, _P, ctypes.POINTER(_i64)]),
    "tdt_dbscan_y_segments": (_i, [_P, _P, _P, _P, _sz, _i, _dbl, _i, _i64, _P, ctypes.POINTER(_i64)]),
    "tdt_dbscan_y_device": (_i, [_P, _P, _P, _sz, ctypes.c_uint64, _i, _i64, _P, _P, ctypes.POINTER(_i)]),
    "tdt_dbscan_device": (_i, [_P, _P, _P, _sz, _P, _i, ctypes.c_uint64, _i, _i, _P, _P]),
    "tdt_sort_dbscan": (_i, [_P, _P, _P, _sz, _P, _i, _dbl, _i, _P, _P]),
    "tdt_sort_dbscan_ex": (_i, [_P, _P, _P, _sz, _P, _i, _dbl, _i, _P, _P, _P, _P]),
    "tdt_cluster_columns": (_i, [_P, _P, _P, _sz, _P, _i, _dbl, _i, _i64, _P, _P, _P]),
    "tdt_host_alloc": (_i, [_sz, _PP]),
    "tdt_host_free": (_i, [_P]),
    "tdt_comm_unique_id": (_i, [_P]),
    "tdt_comm_init": (_i, [_P, _P, _i, _i, _PP]),
    "tdt_comm_destroy": (_i, [_P]),
    "tdt_allgatherv": (_i, [_P, _P, _sz, _P, _P, _P, _i]),
    "tdt_allreduce_sum_f64": (_i, [_P, _P, _sz]),
    "tdt_allgatherv_plan": (_i, [_i, _i, _P, _P, _i, _sz, _P, ctypes.POINTER(_i)]),
    "tdt_signal_select": (_i, [_P, _P, _P, _P, _P, _P, _sz, _P, _i, _i, _i64, _P, ctypes.POINTER(_sz)]),
    "tdt_signal_select_device": (_i, [_P, _P, _P, _P, _P, _P, _sz, _P, _i, _i, _i64, _P, _P]),
    "tdt_signal_scan": (_i, [_P, _P, _sz, _P, _i, _i, _i64, _i, _i, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "tdt_signal_scan_result": (_i, [_P, _P, _P, _P]),
    "tdt_format_clips": (_i, [_P, _P, _P, _P, _sz, ctypes.c_char_p, _P, _sz, ctypes.POINTER(_sz)]),
    "tdt_split_fields": (_i, [_P, _P, _P, _sz, _P, _sz, _i, _P]),
    "tdt_sigtab_create": (_i, [ctypes.c_char_p, _P, _i, _i64, _PP]),
    "tdt_sigtab_destroy": (None, [_P]),
    "tdt_sigtab_add": (_i, [_P, _P, _P, _P, _sz, _sz, _i, _sz, ctypes.POINTER(_sz)]),
    "tdt_sigtab_add_split_row": (_i, [_P, _i, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _P, _i, _i]),
    "tdt_sigtab_add_clips": (_i, [_P, _i, _P, _sz]),
    "tdt_sigtab_clips": (_i, [_P, _i, _PP, ctypes.POINTER(_sz)]),
    "tdt_sigtab_stats": (_i, [_P, _P]),
    "tdt_sigtab_export": (_i, [_P, _P, _i, _P, _sz, ctypes.POINTER(_sz)]),
    "tdt_sigtab_import": (_i, [_P, _P, _sz]),
    "tdt_sigtab_format": (_i, [_P, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "tdt_sigtab_text": (_i, [_P, _i, _PP, ctypes.POINTER(_sz), _P]),
    "tdt_sigtab_sizes": (_i, [_P, _i, _P]),
    "tdt_sigtab_pwrite": (_i, [_P, _i, _i, _i, _i64]),
    "tdt_sigtab_cluster_table": (_i, [_P, _i, _i64, ctypes.POINTER(_sz), ctypes.POINTER(_i)]),
    "tdt_sigtab_cluster_columns": (_i, [_P, _P, _P, _P, _P, _P]),
    "tdt_sigtab_regroup": (_i, [_P, _P, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "tdt_sigtab_regroup_result": (_i, [_P] * 11),
    "tdt_masked_medians": (_i, [_P, _P, _P, _P, _i, _P, _P, _P]),
    "tdt_masked_medians_parts": (_i, [_P, _P, _P, _P, _i, _P, _P, _P]),
    "tdt_segment_means": (_i, [_P, _P, _P, _i64, _P, _P, _P, _sz, _P, _P]),
    "tdt_segment_means_device": (_i, [_P, _P, _P, _P, _P, _P, _sz, _P, _P]),
    "tdt_region_counts": (_i, [_P] * 9 + [_sz, _i, _i64, _P, _P, _P, _sz, _i, _i64, _P]),
    "tdt_region_counts_device": (_i, [_P] * 9 + [_sz, _i, _i, _i64, _P, _P, _P, _sz, _i, _i64, _P]),
    "tdt_format_coverage": (_i, [_P, _sz, ctypes.c_char_p, _i64, _i64, _i, _P, _sz, ctypes.POINTER(_sz)]),
    "tdt_fasta_write_fai": (_i, [ctypes.c_char_p, ctypes.c_char_p]),
    "tdt_host_threads": (_i, [_i]),
    "tdt_bgzf_scan": (_i, [_P, _sz, _sz, _P, _P, _P]),
    "tdt_bgzf_inflate": (_i, [_P, _sz, _P, _sz, _i]),
    "tdt_bgzf_inflate_hbm": (_i, [_P, _P, _sz, _P, _sz, _i]),
    "tdt_ingest_create": (_i, [_P, _i, _PP]),
    "tdt_ingest_destroy": (_i, [_P]),
    "tdt_ingest_push": (_i, [_P, _P, _sz, _sz, ctypes.POINTER(_sz)]),
    "tdt_ingest_push_bounded": (_i, [_P, _P, _sz, _sz, _sz, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "tdt_ingest_prefetch": (_i, [_P, _P, _sz]),
    "tdt_ingest_push_ahead": (_i, [_P, _P, _sz]),
    "tdt_ingest_arrays": (_i, [_P, _PP, ctypes.POINTER(_sz)]),
    "tdt_ingest_packed": (_i, [_P, _PP]),
    "tdt_ingest_bin_for": (_i, [_P, _P, ctypes.POINTER(_i)]),
    "tdt_calib_stream_read": (_i, [_P, _P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "tdt_ingest_timing": (_i, [_P, _P]),
    "tdt_ingest_retain": (_i, [_P, _PP]),
    "tdt_ingest_edge_tids": (_i, [_P, _P, _sz]),
    "tdt_ingest_release": (_i, [_P]),
    "tdt_ingest_edges": (_i, [_P, _P, _sz, ctypes.POINTER(_sz)]),
    "tdt_ingest_carry": (_i, [_P, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "tdt_copy_to_host": (_i, [_P, _P, _P, _sz]),
    "tdt_device_cache_flush": (_i, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    "tdt_device_cache_bytes": (ctypes.c_uint64, [_P]),
    "tdt_debug_fail_next_malloc": (None, [ctypes.c_int]),
    "tdt_stats_create": (_i, [_P, _i64, _i, _i64, _PP]),
    "tdt_stats_destroy": (_i, [_P]),
    "tdt_stats_push_device": (_i, [_P] * 9 + [_sz, ctypes.POINTER(_i)]),
    "tdt_stats_counts": (_i, [_P, _P]),
    "tdt_stats_moments": (_i, [_P, _dbl, _i64, _i64, ctypes.POINTER(_dbl), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "tdt_stats_scan": (_i, [_P] * 8 + [_sz, _i64, _i, _i64, _P, _P, ctypes.POINTER(_sz)]),
    "tdt_bam_decode": (_i, [_P, _sz, _sz, ctypes.POINTER(_sz), ctypes.POINTER(_sz)] + [_P] * 13),
}


def load():
    """dlopen libtiddit_hip.so and type every entry point (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is None:
            # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP/HSA runtimes in
            # one process cannot both own the GPU, so when torch is installed it must be loaded first:
            # libtiddit_hip.so (NEEDED libamdhip64.so.7) then binds to the runtime torch already mapped.
            if "torch" not in sys.modules and not os.environ.get("TIDDIT_AMD_NO_TORCH"):
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            if not os.path.exists(SO_PATH):
                raise ImportError("libtiddit_hip.so is not built (run `python -m tiddit_amd.build`); "
                                  "tiddit_amd has no CPU fallback")
            L = ctypes.CDLL(SO_PATH)
            for name, (res, args) in SYMBOLS.items():
                f = getattr(L, name)
                f.restype = res
                f.argtypes = args
            # a measurement build (ablation / tunable macros, tools/build_variant.sh) is never what the product runs by accident
            flags = L.tdt_build_flags().decode()
            if flags and os.environ.get("TIDDIT_ALLOW_VARIANT") != "1":
                raise ImportError("%s was built with measurement macros (%s); set TIDDIT_ALLOW_VARIANT=1 to load it anyway" % (SO_PATH, flags))
            _lib = L
    return _lib


def check(rc):
    if rc != TDT_OK:
        raise TdtError(rc, load().tdt_last_error().decode(errors="replace"))


def ptr(a):
    """host numpy array -> void*"""
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Context:
    """One tdt_ctx (device + stream + scratch).  Contexts are per process; use default_context()."""

    def __init__(self, device=0):
        self.lib = load()
        h = ctypes.c_void_p()
        check(self.lib.tdt_ctx_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device

    def sync(self):
        check(self.lib.tdt_ctx_sync(self.handle))

    def set_stream(self, hip_stream):
        check(self.lib.tdt_ctx_set_stream(self.handle, ctypes.c_void_p(hip_stream)))

    def close(self):
        if self.handle:
            self.lib.tdt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("TIDDIT_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (os.getpid(), device)
    if key not in _default:
        _default[key] = Context(device)
    return _default[key]
