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


def _device_figures(lib, h, state):
    """counters and the three numpy figures of the insert-size list from the device (csrc/tdt_stats.hip); None without insert sizes"""
    cnt = numpy.zeros(9, dtype=numpy.int64)
    _native.check(lib.tdt_stats_counts(h, _native.ptr(cnt)))
    state[:5] = cnt[:5]
    n_ins = int(cnt[5])
    if not n_ins:
        return None
    mean = numpy.float64(int(cnt[6])) / n_ins                       # numpy.average of ints: an exact float64 sum / n
    q = numpy.true_divide(99.9, 100)                                # numpy.percentile's own arithmetic (method "linear")
    vi = (n_ins - 1) * q
    k0 = int(numpy.floor(vi))
    k1 = min(k0 + 1, n_ins - 1)
    msd, o0, o1 = ctypes.c_double(0), ctypes.c_int32(0), ctypes.c_int32(0)
    _native.check(lib.tdt_stats_moments(h, float(mean), k0, k1, ctypes.byref(msd), ctypes.byref(o0), ctypes.byref(o1)))
    a, b, g = numpy.int32(o0.value), numpy.int32(o1.value), vi - k0
    diff = numpy.subtract(b, a)
    pct = numpy.add(a, diff * g)                                    # numpy's _lerp(a, b, t)
    if g >= 0.5:
        pct = numpy.subtract(b, diff * (1 - g))
    return mean, numpy.sqrt(numpy.float64(msd.value)), numpy.float64(pct)


STAGE_SECONDS = {}          # wall seconds inside the last statistics() call, by what it waited for


def _statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads, carry=False, shard=None):
    import os
    from . import bamio
    library = {}
    t = time.time()
    STAGE_SECONDS.clear()
    bamio.set_carry(None)
    STAGE_SECONDS["drop a stale carry"] = time.time() - t
    # shard = (0, world): rank 0 of an N-rank job samples from ITS byte range of the file, so that the batches it keeps are the head of
    # its own scan; a sample that does not end inside that range is taken again from the whole file (below)
    sharded = shard is not None and os.environ.get("TIDDIT_HOST_INGEST") != "1" and os.environ.get("TIDDIT_STATS_HOST") != "1"
    reader = bamio.DeviceBamReader(bam_file_name, shard=shard, chunk=int(os.environ.get("TIDDIT_INGEST_CHUNK", str(448 << 20)))) if sharded \
        else open_bam(bam_file_name)
    # `tiddit --sv` scans the same file for signals next (tiddit_signal.main): the sampled batches stay in HBM with their coverage records
    # written for the 50-bp histogram, and that pass starts from them instead of reading and inflating this part of the file again
    STAGE_SECONDS["open the reader"] = time.time() - t
    on_device = isinstance(reader, bamio.DeviceBamReader)
    carry = carry and on_device and os.environ.get("TIDDIT_NO_CARRY") != "1"
    kept, hist = [], None
    if carry:
        from . import tiddit_coverage
        hist = tiddit_coverage.CoverageHistogram([(n, l) for n, l in zip(reader.references, reader.lengths)], 50, ctx=reader.ctx)
        reader.bin_for(hist)
        reader.retain = True
    STAGE_SECONDS["50-bp histogram for the carried batches"] = time.time() - t - STAGE_SECONDS["open the reader"]
    lib = _native.load()
    state = numpy.zeros(6, dtype=numpy.int64)
    chunks = []
    figures = None
    batches = reader.batches()
    t_loop, t_first = time.time(), None
    if on_device and os.environ.get("TIDDIT_STATS_HOST") != "1":
        # the sampling loop (:17-47), its cut-off and the three numpy figures (:52-56) on the device: the decoded fields of every batch
        # are in HBM already, and nothing but a handful of counters comes back
        h = ctypes.c_void_p()
        _native.check(lib.tdt_stats_create(reader.ctx.handle, int(n_reads), int(min_mapq), int(max_ins_len), ctypes.byref(h)))
        try:
            done = ctypes.c_int(0)
            for b in batches:
                if t_first is None:
                    t_first = time.time()
                    STAGE_SECONDS["first batch (pinned spans, first read, first push)"] = t_first - t_loop
                if carry:
                    kept.append(b)
                d = b.dev
                _native.check(lib.tdt_stats_push_device(h, d["tid"], d["pos"], d["mate_tid"], d["mate_pos"], d["tlen"], d["l_seq"], d["flag"], d["mapq"],
                                                        len(b), ctypes.byref(done)))
                if done.value:
                    break
            if sharded and not done.value:
                # the share ended before the sample did (a small file, many ranks): the reference samples on into the rest of the file
                lib.tdt_stats_destroy(h)
                h = None
                for b in kept:
                    b.release()
                batches.close()
                reader.close()
                if hist is not None:
                    hist.close()
                return _statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads, carry=False, shard=None)
            STAGE_SECONDS["the other batches + sampling kernels"] = time.time() - (t_first or t_loop)
            t_f = time.time()
            figures = _device_figures(lib, h, state)
            STAGE_SECONDS["figures (mean, std, percentile)"] = time.time() - t_f
        finally:
            if h is not None:
                lib.tdt_stats_destroy(h)
    else:
        def scan(cols, n):
            # the sampling loop of the reference (:17-47), read by read, in C (csrc/tdt_bam.hip: tdt_stats_scan); `state` carries over
            out = numpy.empty(n, dtype=numpy.int32)
            k = ctypes.c_size_t(0)
            _native.check(lib.tdt_stats_scan(*[_native.ptr(c) for c in cols], n, int(n_reads), int(min_mapq), int(max_ins_len), _native.ptr(state),
                                             _native.ptr(out), ctypes.byref(k)))
            chunks.append(out[:k.value].copy())

        # one worker runs the loop of batch k (ctypes drops the GIL) while batch k + 1 is inflated and decoded; batches are scanned in
        # order, and the pass stops one batch after the loop has seen its n_reads-th read
        pending = None
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
        insert_size = numpy.concatenate(chunks) if chunks else numpy.zeros(0, dtype=numpy.int32)
        if len(insert_size):
            figures = (numpy.average(insert_size), numpy.std(insert_size), numpy.percentile(insert_size, 99.9))
    if carry:
        reader.retain = False
        bamio.set_carry(bamio.ScanCarry(bam_file_name, reader, batches, kept, hist))
    else:
        batches.close()                          # (the generator's finally stops the span thread and hands its pinned buffers back)
        reader.close()
    is_innie, is_outtie = int(state[3]), int(state[4])
    # numpy.average of the read lengths: an exact integer sum over an exact count
    avg_read_length = (float(state[1]) / float(state[2])) if state[2] else float(numpy.average(numpy.zeros(0)))
    library["avg_read_length"] = avg_read_length
    if figures is not None:
        library["avg_insert_size"], library["std_insert_size"], library["percentile_insert_size"] = figures
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


def statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads, carry=False, shard=None):
    """``tiddit_stats.statistics`` (tiddit_stats.py:5-78); the collector is off meanwhile (hostutil.quiet_gc).
    carry=True (the one-process `tiddit --sv` sets it: the signal scan of the same file follows at once and takes it,
    bamio.take_carry): the sampled batches stay in HBM with an open reader for that scan.  A library caller leaves nothing behind.
    shard=(0, world): rank 0 of an N-rank job — the sample is read through the reader of rank 0's byte range, so the carry is the head
    of that rank's own scan (same figures: the sample is a prefix of the file either way)."""
    with quiet_gc():
        return _statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads, carry=carry, shard=shard)
