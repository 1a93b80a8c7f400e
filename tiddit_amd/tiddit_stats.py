"""Drop-in for ``tiddit.tiddit_stats.statistics`` (tiddit_stats.py:5-78): library statistics from the
first ``n_reads`` placed alignments — mean read length, insert-size mean / std / 99.9th percentile and
the pair-orientation vote.  Same sampling rules, evaluated on the decoded arrays instead of per read."""
import concurrent.futures
import time

import ctypes

import numpy

from .hostutil import quiet_gc
from . import _native
from .bamio import open_bam


def _statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads):
    library = {}
    t = time.time()
    import os
    from . import bamio
    bamio.set_carry(None)
    reader = open_bam(bam_file_name)
    # `tiddit --sv` scans the same file for signals next (tiddit_signal.main): the sampled batches stay in HBM with their coverage records
    # written for the 50-bp histogram, and that pass starts from them instead of reading and inflating this part of the file again
    carry = isinstance(reader, bamio.DeviceBamReader) and os.environ.get("TIDDIT_NO_CARRY") != "1"
    kept, hist = [], None
    if carry:
        from . import tiddit_coverage
        hist = tiddit_coverage.CoverageHistogram([(n, l) for n, l in zip(reader.references, reader.lengths)], 50, ctx=reader.ctx)
        reader.bin_for(hist)
        reader.retain = True
    lib = _native.load()
    state = numpy.zeros(6, dtype=numpy.int64)
    chunks = []

    def scan(cols, n):
        # the sampling loop of the reference (:17-47), read by read, in C (csrc/tdt_bam.hip: tdt_stats_scan); `state` carries over
        out = numpy.empty(n, dtype=numpy.int32)
        k = ctypes.c_size_t(0)
        _native.check(lib.tdt_stats_scan(*[_native.ptr(c) for c in cols], n, int(n_reads), int(min_mapq), int(max_ins_len), _native.ptr(state),
                                         _native.ptr(out), ctypes.byref(k)))
        chunks.append(out[:k.value].copy())

    # one worker runs the loop of batch k (ctypes drops the GIL) while the device inflates and decodes batch k + 1; batches are
    # scanned in order, and the pass stops one batch after the loop has seen its n_reads-th read
    pending = None
    batches = reader.batches()
    with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
        for b in batches:
            if carry:
                kept.append(b)
            cols = [numpy.ascontiguousarray(getattr(b, k)) for k in ("tid", "pos", "mate_tid", "mate_pos", "tlen", "l_seq", "flag", "mapq")]
            if pending is not None:
                pending.result()
                if state[5]:
                    pending = None
                    break
            pending = pool.submit(scan, cols, len(b))
        if pending is not None:
            pending.result()
    if carry:
        reader.retain = False
        bamio.set_carry(bamio.ScanCarry(bam_file_name, reader, batches, kept, hist))
    else:
        reader.close()
    insert_size = numpy.concatenate(chunks) if chunks else numpy.zeros(0, dtype=numpy.int32)
    is_innie, is_outtie = int(state[3]), int(state[4])
    # numpy.average of the read lengths: an exact integer sum over an exact count
    avg_read_length = (float(state[1]) / float(state[2])) if state[2] else float(numpy.average(numpy.zeros(0)))
    library["avg_read_length"] = avg_read_length
    if len(insert_size):
        library["avg_insert_size"] = numpy.average(insert_size)
        library["std_insert_size"] = numpy.std(insert_size)
        library["percentile_insert_size"] = numpy.percentile(insert_size, 99.9)
    else:
        library["avg_insert_size"] = 0
        library["std_insert_size"] = 0
        library["percentile_insert_size"] = 0
    print("LIBRARY STATISTICS")
    if is_innie > is_outtie:
        library["mp"] = False
        print("\tPair orientation = Forward-Reverse")
    else:
        print("\tPair orientation = Reverse-Forward")
        library["mp"] = True
    print("\tAverage Read length = {}".format(library["avg_read_length"]))
    print("\tAverage insert size = {}".format(library["avg_insert_size"]))
    print("\tStdev insert size = {}".format(library["std_insert_size"]))
    print("\t99.95 percentile insert size = {}".format(library["percentile_insert_size"]))
    print("Calculated statistics in: " + str(t - time.time()))
    print("")
    return library


def statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads):
    """``tiddit_stats.statistics`` (tiddit_stats.py:5-78); the collector is off meanwhile (hostutil.quiet_gc)."""
    with quiet_gc():
        return _statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads)
