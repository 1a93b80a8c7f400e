"""Drop-in for ``tiddit.tiddit_coverage`` (tiddit_coverage.pyx) on the MI355X.

Same entry points as the reference — ``create_coverage`` (:10-21), ``update_coverage`` (:48-74),
``print_coverage`` (:22-45) — plus the batch forms the per-read loops of ``__main__.py:229-242`` and
``tiddit_signal.pyx:169-182`` are rewritten onto: ``update_coverage_batch`` and ``CoverageHistogram``.
Bins are bit-identical to the reference's float64 arrays (see csrc/tdt_coverage.hip).
"""
import ctypes
import threading
import math

import numpy

from . import _native


def create_coverage(bam_header, bin_size, c="all"):
    """tiddit_coverage.pyx:10-21 — zeroed float64 bins per contig + the size of each last bin."""
    coverage_data = {}
    end_bin_size = {}
    for contig in bam_header["SQ"]:
        if c == "all" or contig["SN"] == c:
            bins = int(math.ceil(contig["LN"] / float(bin_size)))
            coverage_data[contig["SN"]] = numpy.zeros(bins)
            end_bin_size[contig["SN"]] = contig["LN"] - (bins - 1) * bin_size
            if c != "all":
                return (coverage_data[contig["SN"]], end_bin_size[contig["SN"]])
    return (coverage_data, end_bin_size)


def print_coverage(coverage_data, bam_header, bin_size, file_type, outfile):
    """tiddit_coverage.pyx:22-45 — bed (note the reference's `+1` bin end and LN on the last row)
    or fixedStep wig; values are formatted exactly like ``"{}".format(numpy.float64)``."""
    lib = _native.load()
    f = open(outfile, "wb")                     # buffered: a raw FileIO.write may write short and does not retry
    if file_type == "bed":
        f.write(b"#chromosome\tstart\tend\tcoverage\n")
    elif file_type == "wig":
        f.write(b"track type=wiggle_0 name=\"Coverage\" description=\"Per bin average coverage\"\n")
    for contig in bam_header["SQ"]:
        name = contig["SN"]
        values = numpy.ascontiguousarray(coverage_data[name], dtype=numpy.float64)
        if file_type == "wig":
            f.write("fixedStep chrom={} start=1 step={}\n".format(name, bin_size).encode())
        elif file_type != "bed":
            continue
        if len(name.encode()) > 100 or not len(values):                 # odd names: the literal loop of the reference
            f.write(_rows_python(values, name, contig["LN"], bin_size, file_type).encode())
            continue
        # the row loop (:34-44) on the host thread pool, byte-identical to "{}".format(numpy.float64)
        kind = 0 if file_type == "bed" else 1
        size = ctypes.c_size_t(0)
        cap = len(values) * (26 if kind else 50 + len(name.encode())) + 64
        buf = numpy.empty(cap, dtype=numpy.uint8)
        _native.check(lib.tdt_format_coverage(_native.ptr(values), len(values), name.encode(), int(bin_size), int(contig["LN"]), kind,
                                              _native.ptr(buf), cap, ctypes.byref(size)))
        f.write(memoryview(buf)[:size.value])
    f.close()


def _rows_python(values, name, contig_len, bin_size, file_type):
    n = len(values)
    if file_type == "wig":
        return "".join("{}\n".format(v) for v in values)
    rows = []
    for i in range(0, n):
        bin_end = (i + 1) * bin_size + 1
        if i == n - 1:
            bin_end = contig_len
        rows.append("{}\t{}\t{}\t{}\n".format(name, 1 + i * bin_size, bin_end, values[i]))
    return "".join(rows)


class CoverageHistogram:
    """Device-resident binned read-depth histogram over a set of contigs.

    ``push(contig, start, end, mapq, flag, min_q)`` adds a batch of alignment records (0-based
    start, exclusive end, the read filter of __main__.py:231-235 / tiddit_signal.pyx:171-181 is
    applied on device); ``finish(contig)`` returns the float64 bins.
    """

    def __init__(self, contigs, bin_size, ctx=None):
        """contigs: list of (name, length) or a bam header dict."""
        if isinstance(contigs, dict):
            contigs = [(c["SN"], c["LN"]) for c in contigs["SQ"]]
        self.ctx = ctx or _native.default_context()
        self.lib = self.ctx.lib
        self.names = [c[0] for c in contigs]
        self.lengths = numpy.array([c[1] for c in contigs], dtype=numpy.int64)
        self.tid = {n: i for i, n in enumerate(self.names)}
        self.bin_size = int(bin_size)
        h = ctypes.c_void_p()
        _native.check(self.lib.tdt_cov_create(self.ctx.handle, _native.ptr(self.lengths), len(self.names), self.bin_size,
                                              ctypes.byref(h)))
        self.handle = h

    def nbins(self, contig):
        nb = ctypes.c_int64()
        eb = ctypes.c_int()
        _native.check(self.lib.tdt_cov_nbins(self.handle, self._tid(contig), ctypes.byref(nb), ctypes.byref(eb)))
        return nb.value, eb.value

    def _tid(self, contig):
        return contig if isinstance(contig, (int, numpy.integer)) else self.tid[contig]

    def push(self, contig, start, end, mapq, flag, min_q):
        start = numpy.ascontiguousarray(start, dtype=numpy.int32)
        end = numpy.ascontiguousarray(end, dtype=numpy.int32)
        mapq = numpy.ascontiguousarray(mapq, dtype=numpy.uint8)
        flag = numpy.ascontiguousarray(flag, dtype=numpy.uint16)
        n = len(start)
        if not (len(end) == len(mapq) == len(flag) == n):
            raise ValueError("start/end/mapq/flag must have the same length")
        _native.check(self.lib.tdt_cov_push(self.handle, self._tid(contig), _native.ptr(start), _native.ptr(end),
                                            _native.ptr(mapq), _native.ptr(flag), n, int(min_q)))

    def push_device(self, contig, d_start, d_end, d_mapq, d_flag, n, min_q):
        """Device pointers (ints), e.g. ``tensor.data_ptr()``; asynchronous on the context stream."""
        _native.check(self.lib.tdt_cov_push_device(self.handle, self._tid(contig), d_start, d_end, d_mapq, d_flag, n, int(min_q)))

    def push_device_multi(self, items, min_q):
        """items: list of (contig, d_start, d_end, d_mapq, d_flag, n) with device pointers — ONE launch."""
        k = len(items)
        tids = numpy.array([self._tid(it[0]) for it in items], dtype=numpy.int32)
        ptrs = [numpy.array([it[j] for it in items], dtype=numpy.uint64) for j in (1, 2, 3, 4)]
        ns = numpy.array([it[5] for it in items], dtype=numpy.uint64)
        _native.check(self.lib.tdt_cov_push_device_multi(self.handle, k, _native.ptr(tids), _native.ptr(ptrs[0]), _native.ptr(ptrs[1]),
                                                         _native.ptr(ptrs[2]), _native.ptr(ptrs[3]), _native.ptr(ns), int(min_q)))

    def push_packed_device_multi(self, items, min_q):
        """items: list of (contig, d_packed, d_end or 0, n): 8-byte packed records (tdt_cov_pack_device / the ingest kernel) — ONE launch."""
        k = len(items)
        tids = numpy.array([self._tid(it[0]) for it in items], dtype=numpy.int32)
        pk = numpy.array([it[1] for it in items], dtype=numpy.uint64)
        en = numpy.array([it[2] or 0 for it in items], dtype=numpy.uint64)
        ns = numpy.array([it[3] for it in items], dtype=numpy.uint64)
        _native.check(self.lib.tdt_cov_push_packed_device_multi(self.handle, k, _native.ptr(tids), _native.ptr(pk), _native.ptr(en), _native.ptr(ns),
                                                                int(min_q)))

    def has_binned(self):
        """binned records (csrc/tdt_common.h: cov_bin_record) exist for 2 <= bin_size < 1024"""
        return 2 <= self.bin_size < 1024

    def pack_binned_device(self, contig, d_start, d_end, d_mapq, d_flag, n, d_out):
        """four device arrays of one contig -> n binned records for THIS histogram's bin size at d_out (8 B each)"""
        _native.check(self.lib.tdt_cov_pack_binned_device(self.handle, self._tid(contig), d_start, d_end, d_mapq, d_flag, n, d_out))

    def push_binned_device_multi(self, items, min_q):
        """items: list of (contig, d_binned, d_start, d_end, n): 8-byte binned records of this histogram's bin size (pack_binned_device /
        a DeviceBamReader bound with ``bin_for``) plus the start / end arrays the literal path replays long reads from — ONE launch."""
        k = len(items)
        tids = numpy.array([self._tid(it[0]) for it in items], dtype=numpy.int32)
        pk = numpy.array([it[1] for it in items], dtype=numpy.uint64)
        st = numpy.array([it[2] for it in items], dtype=numpy.uint64)
        en = numpy.array([it[3] for it in items], dtype=numpy.uint64)
        ns = numpy.array([it[4] for it in items], dtype=numpy.uint64)
        _native.check(self.lib.tdt_cov_push_binned_device_multi(self.handle, k, _native.ptr(tids), _native.ptr(pk), _native.ptr(st), _native.ptr(en),
                                                                _native.ptr(ns), int(min_q)))

    def push_device_batch(self, batch, min_q, want=None):
        """every per-contig run of a DeviceBatch (bamio.DeviceBamReader) in ONE launch, through the 8-byte packed records the
        ingest kernel wrote when min_q fits their 6-bit mapq field; want[tid] false skips a contig"""
        d = batch.dev
        runs = [(t, lo, hi) for t, lo, hi in batch.runs if t >= 0 and (want is None or want[t])]
        if not runs:
            return
        if d.get("packed") and int(min_q) <= 63 and getattr(batch, "binned_for", None) is self:
            self.push_binned_device_multi([(t, d["packed"] + 8 * lo, d["pos"] + 4 * lo, d["end"] + 4 * lo, hi - lo) for t, lo, hi in runs], min_q)
        elif d.get("packed") and int(min_q) <= 63 and getattr(batch, "binned_for", None) is None:
            self.push_packed_device_multi([(t, d["packed"] + 8 * lo, d["end"] + 4 * lo, hi - lo) for t, lo, hi in runs], min_q)
        else:
            self.push_device_multi([(t, d["pos"] + 4 * lo, d["end"] + 4 * lo, d["mapq"] + lo, d["flag"] + 2 * lo, hi - lo) for t, lo, hi in runs], min_q)

    def total_bins(self):
        t = ctypes.c_int64()
        _native.check(self.lib.tdt_cov_total_bins(self.handle, ctypes.byref(t)))
        return t.value

    def offset(self, contig):
        o = ctypes.c_int64()
        _native.check(self.lib.tdt_cov_offset(self.handle, self._tid(contig), ctypes.byref(o)))
        return o.value

    def finish_all_device(self, d_out):
        """float64[total_bins()] on the device; contig c occupies [offset(c), offset(c)+nbins(c))."""
        _native.check(self.lib.tdt_cov_finish_all_device(self.handle, d_out))

    def finish_all(self):
        """float64[total_bins()] on the host; contig c occupies [offset(c), offset(c) + nbins(c))"""
        out = numpy.empty(max(1, self.total_bins()), dtype=numpy.float64)
        _native.check(self.lib.tdt_cov_finish_all(self.handle, _native.ptr(out)))
        return out[:self.total_bins()]

    def finish(self, contig):
        nb, _ = self.nbins(contig)
        out = numpy.empty(nb, dtype=numpy.float64)
        _native.check(self.lib.tdt_cov_finish(self.handle, self._tid(contig), _native.ptr(out)))
        return out

    def finish_device(self, contig, d_out):
        _native.check(self.lib.tdt_cov_finish_device(self.handle, self._tid(contig), d_out))

    def kept(self):
        k = ctypes.c_int64()
        _native.check(self.lib.tdt_cov_kept(self.handle, ctypes.byref(k)))
        return k.value

    def reset(self):
        _native.check(self.lib.tdt_cov_reset(self.handle))

    def close(self):
        if self.handle:
            self.lib.tdt_cov_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _contig_length(coverage_data, bin_size, end_bin_size):
    return (len(coverage_data) - 1) * bin_size + end_bin_size if len(coverage_data) else 0


def update_coverage_batch(ref_start, ref_end, mapq, flag, min_q, bin_size, coverage_data, end_bin_size):
    """Batch form of the per-read loop: filter + update_coverage for every record, in place.

    Equivalent to calling the reference's ``update_coverage`` for each kept read in any order
    (the float64 sum is exact).  Returns ``coverage_data`` like the reference.
    """
    LN = _contig_length(coverage_data, bin_size, end_bin_size)
    key = (LN, int(bin_size))
    with _HIST_LOCK:                  # the cached histogram is shared state: reset / push / finish of two callers must not interleave
        h = _HIST_CACHE.get(key)      # one histogram object per (contig length, bin size): a per-read loop reuses it
        if h is None:
            if len(_HIST_CACHE) >= 8:
                _HIST_CACHE.pop(next(iter(_HIST_CACHE))).close()
            h = _HIST_CACHE[key] = CoverageHistogram([("c", LN)], bin_size)
        try:
            h.reset()
            h.push(0, ref_start, ref_end, mapq, flag, min_q)
            coverage_data += h.finish(0)
        except Exception:
            _HIST_CACHE.pop(key, None)
            h.close()
            raise
    return coverage_data


def clear_histogram_cache():
    """release the device histograms `update_coverage[_batch]` keeps between calls"""
    with _HIST_LOCK:
        while _HIST_CACHE:
            _HIST_CACHE.popitem()[1].close()


_HIST_CACHE = {}
_HIST_LOCK = threading.RLock()


def update_coverage(ref_start, ref_end, bin_size, coverage_data, end_bin_size):
    """tiddit_coverage.pyx:48-74 — one read.  Kept for drop-in compatibility (one device round trip
    per call: use ``update_coverage_batch`` / ``CoverageHistogram`` in loops).  Raises IndexError where
    the reference does (bin index outside the array)."""
    try:
        return update_coverage_batch(numpy.array([ref_start]), numpy.array([ref_end]), numpy.array([255], dtype=numpy.uint8),
                                     numpy.array([0], dtype=numpy.uint16), 0, bin_size, coverage_data, end_bin_size)
    except _native.TdtError as e:
        if e.code == -3:
            raise IndexError("index out of bounds for coverage_data") from e
        raise
