"""Drop-in for ``tiddit.tiddit_signal`` (tiddit_signal.pyx) on the MI355X.

``main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index,
min_anchor_len, min_clip_len) -> coverage_data`` (:230-334) with the same side-effect files
(``{prefix}_tiddit/discordants_{sample}.tab``, ``splits_{sample}.tab``, ``clips_{sample}.fa``,
``clips/{contig}.fa``), plus ``SA_analysis`` / ``find_SA_query_range`` (:11-145).

Redesign: the reference forks one joblib worker per contig and calls Python once per read
(``worker`` :147-228).  Here ONE process streams the BAM through the C record decoder
(``bamio.BamReader``); the packed start/end/mapq/flag arrays of every batch go to the device coverage
histogram (bin 50, same read filter, :171-182), and the signal predicates (:184-221) are evaluated on
whole arrays on the device (``tdt_signal_scan``).  The ~1 % of reads that are discordant, split or clipped go from the
batch into NATIVE signal tables (``sigtab.SignalTables``, csrc/tdt_sigtab.hip): merged per (chrA, chrB, fragment) while the file is
still being scanned, formatted as the .tab / clip files by the host thread pool, and handed to ``tiddit_cluster`` in the same
process — no Python object per read anywhere.  The literal Python merge (``_merge_and_write``) is what the host-ingest mode
(``TIDDIT_HOST_INGEST=1``) runs, and what the native tables are tested against.
``threads`` and ``skip_index`` are accepted for signature compatibility and ignored (no index needed).
"""
import concurrent.futures
import itertools
import os
import re
import threading
import time

import numpy

import ctypes

from .hostutil import quiet_gc
from . import _native, tiddit_coverage
from .bamio import DeviceBamReader, DeviceBatch, RecordView, open_bam

_SA_OPS = {"M": 0, "S": 4, "H": 5, "D": 2, "I": 1}   # :23 — any other CIGAR letter raises KeyError, like the reference


class _SASegment:
    """The synthetic AlignedSegment the reference builds for an SA entry (:11-29), reduced to the
    attributes SA_analysis reads.  reference_start is the SA tag's 1-based POS stored raw (:13)."""

    def __init__(self, pos, is_minus, cigar):
        self.reference_start = pos
        self.flag = 80 if is_minus else 64
        self.cigartuples = cigar
        ref = sum(l for op, l in cigar if op in (0, 2))
        self.reference_end = pos + (ref if ref else 1)               # htslib bam_endpos
        self.query_alignment_start = _leading_softclip(cigar)
        # no sequence: pysam derives the end from the CIGAR (leading S + M/I lengths)
        self.query_alignment_end = self.query_alignment_start + sum(l for op, l in cigar if op in (0, 1))


def _leading_softclip(cigar):
    n = 0
    for op, l in cigar:
        if op == 5:
            continue
        if op == 4:
            n += l
        else:
            break
    return n


_CIGAR_OK = re.compile(r"(?:[0-9]+[^0-9])+\Z")          # ASCII digit runs each followed by ONE other character: the well-formed case
_CIGAR_PAIR = re.compile(r"([0-9]+)([^0-9])")


def find_SA_query_range(SA):
    """SA = one entry split on ',': rname,pos,strand,CIGAR,mapQ,NM  (:11-29)"""
    if _CIGAR_OK.match(SA[3]):                          # same pairs as the grouping below, without building the groups
        cigar = [(_SA_OPS[op], int(n)) for n, op in _CIGAR_PAIR.findall(SA[3])]
    else:
        parts = ["".join(x) for _, x in itertools.groupby(SA[3], key=str.isdigit)]
        cigar = [(_SA_OPS[parts[i * 2 + 1]], int(parts[i * 2])) for i in range(0, int(len(parts) / 2))]
    return _SASegment(int(SA[1]), SA[2] != "+", cigar)


def SA_analysis(read, min_q, tag, reference_name):
    """Split-read signal of a read carrying an SA tag (:31-145).  ``read`` needs query_name,
    reference_start, reference_end, is_reverse, query_alignment_start and get_tag(tag).
    QUIRK (:36-39): with several SA entries the selection loop only ever inspects entry 0, so entry 0
    is always the one used and the read is dropped when its mapQ is below min_q."""
    entries = read.get_tag(tag).rstrip(";").split(";")
    sa = entries[0].split(",")
    if int(sa[4]) < min_q:
        return ()
    seg = find_SA_query_range(sa)
    clip_before = seg.query_alignment_start < read.query_alignment_start
    read_start, read_end = read.reference_start + 1, read.reference_end + 1
    if clip_before:
        split_pos = read_end if read.is_reverse else read_start
    else:
        split_pos = read_start if read.is_reverse else read_end
    sa_minus = sa[2] == "-"
    if clip_before:
        sa_split = seg.reference_start if sa_minus else seg.reference_end
    else:
        sa_split = seg.reference_end if sa_minus else seg.reference_start
    sa_chr = sa[0]
    startA, endA, startB, endB = read_start, read_end, seg.reference_start, seg.reference_end
    swap = False
    if sa_chr < reference_name:                     # string order, like the reference
        chrA, chrB, swap = sa_chr, reference_name, True
    else:
        chrA, chrB = reference_name, sa_chr
        if chrA == chrB and sa_split < split_pos:
            swap = True
    if swap:
        split_pos, sa_split = sa_split, split_pos
        startA, endA, startB, endB = seg.reference_start, seg.reference_end, read_start, read_end
    return [chrA, chrB, read.query_name, split_pos, read.is_reverse, sa_split, sa_minus, startA, endA, startB, endB]


class _ReadProxy:
    """What SA_analysis needs from a decoded record."""

    def __init__(self, batch, i):
        self._rec = batch.record(i)
        self.reference_start = int(batch.pos[i])
        self.reference_end = int(batch.end[i])
        self.is_reverse = bool(batch.flag[i] & 0x10)

    @property
    def query_name(self):
        return self._rec.query_name

    @property
    def query_alignment_start(self):
        return _leading_softclip(self._rec.cigartuples)

    def get_tag(self, tag):
        return self._rec.get_tag_sa()


_META = numpy.dtype([("idx", "<u4"), ("tid", "<i4"), ("pos", "<i4"), ("end", "<i4"), ("mate_tid", "<i4"), ("sa_rel", "<i4"),
                     ("flag", "<u2"), ("action", "u1"), ("pad", "u1")])
_SPLIT = numpy.dtype([("status", "<i4"), ("read_start", "<i4"), ("read_end", "<i4"), ("split_pos", "<i4"), ("sa_split", "<i4"), ("seg_start", "<i4"),
                      ("seg_end", "<i4"), ("chr_off", "<u4"), ("chr_len", "<u4"), ("is_reverse", "u1"), ("sa_minus", "u1"), ("pad", "<u2")])
_FIELD_ORDER = ("tid", "pos", "end", "mapq", "flag", "mate_tid", "mate_pos", "tlen", "l_seq", "cigar_first", "cigar_last", "rec_off", "sa_off", "raw")


class SelectedReads:
    """The reads of one device batch that tiddit_signal.worker acts on (clip / split / discordant), gathered by
    ``tdt_signal_scan``: field arrays plus their raw records back to back.  Quacks like a decoded batch for RecordView."""

    def __init__(self, ctx, meta, raw_end, raw):
        self.ctx, self.meta, self.raw_end, self.raw = ctx, meta, raw_end, raw
        self.tid, self.pos, self.end, self.flag, self.mate_tid, self.action = (meta[k] for k in ("tid", "pos", "end", "flag", "mate_tid", "action"))

    # What only the per-record view needs (RecordView: the literal SA_analysis of an unusual tag, worker()'s row lists) is made when it
    # is first asked for — the scan itself hands meta / raw_end / raw to the native tables as they are, and a bytes copy of every
    # batch's selected records (tens of MB) sat on the scanning thread's path for nothing.
    @property
    def raw_bytes(self):
        v = self.__dict__.get("_raw_bytes")
        if v is None:
            v = self.__dict__["_raw_bytes"] = self.raw.tobytes()      # RecordView slices this copy
        return v

    @property
    def rec_off(self):
        v = self.__dict__.get("_rec_off")
        if v is None:
            raw_end = self.raw_end
            v = numpy.concatenate([[0], raw_end[:-1]]).astype(numpy.uint64) if len(raw_end) else numpy.zeros(0, dtype=numpy.uint64)
            self.__dict__["_rec_off"] = v
        return v

    @property
    def sa_off(self):
        v = self.__dict__.get("_sa_off")
        if v is None:
            v = self.__dict__["_sa_off"] = numpy.where(self.meta["sa_rel"] >= 0, self.rec_off.astype(numpy.int64) + self.meta["sa_rel"], -1)
        return v

    def __len__(self):
        return len(self.meta)

    def record(self, k):
        return RecordView(self, k)

    def clip_fasta(self, which, contig):
        """``>name|contig|pos+1\\nSEQ\\n`` of the selected records `which`, as one bytes object (tdt_format_clips)"""
        which = numpy.ascontiguousarray(which, dtype=numpy.uint32)
        lib = self.ctx.lib
        need = ctypes.c_size_t(0)
        args = (_native.ptr(self.meta), _native.ptr(self.raw_end), _native.ptr(self.raw), _native.ptr(which), len(which), contig.encode())
        _native.check(lib.tdt_format_clips(*args, None, 0, ctypes.byref(need)))
        buf = numpy.empty(need.value, dtype=numpy.uint8)
        _native.check(lib.tdt_format_clips(*args, _native.ptr(buf), need.value, ctypes.byref(need)))
        return buf[:need.value].tobytes()


def _clip_bytes(clip):
    """the bytes of one clip entry: [header, sequence] strings of the host path, or [bytes, ""] of a whole batch formatted in C"""
    return clip[0] if isinstance(clip[0], bytes) else "".join(clip).encode()


def split_rows_native(sel, which4, names, min_q, splits, lib=None):
    """split-read rows (SA_analysis, :31-145) of the selected reads `which4` appended to splits[contig]: the numeric half in C
    (``tdt_split_fields``) for well-formed tags; contig names, their string order and the swap (:118-140) here; unusual tags go
    through the literal :func:`SA_analysis` (and raise what it raises)."""
    if not len(which4):
        return
    lib = lib or sel.ctx.lib
    so = numpy.empty(len(which4), dtype=_SPLIT)
    _native.check(lib.tdt_split_fields(_native.ptr(sel.meta), _native.ptr(sel.raw_end), _native.ptr(sel.raw), len(sel.raw),
                                       _native.ptr(numpy.ascontiguousarray(which4, dtype=numpy.uint32)), len(which4), int(min_q), _native.ptr(so)))
    rb = sel.raw_bytes
    cols = zip(which4.tolist(), so["status"].tolist(), so["read_start"].tolist(), so["read_end"].tolist(), so["split_pos"].tolist(),
               so["sa_split"].tolist(), so["seg_start"].tolist(), so["seg_end"].tolist(), so["chr_off"].tolist(), so["chr_len"].tolist(),
               so["is_reverse"].tolist(), so["sa_minus"].tolist(), sel.tid[which4].tolist(), sel.rec_off[which4].tolist())
    for k, status, rs, re_, sp, ssp, gs, ge, co, cl, rev, sam, t_, o_ in cols:
        if status == 0:
            continue
        chrom = names[t_]
        if status != 1:                                   # an unusual tag: the literal code (and its exceptions)
            split = SA_analysis(_ReadProxy(sel, k), min_q, "SA", chrom)
            if split:
                splits[chrom].append(split)
            continue
        sa_chr = rb[co:co + cl].decode()
        qname = rb[o_ + 36:o_ + 35 + rb[o_ + 12]].decode()
        if sa_chr < chrom:                                # string order, like the reference (:118)
            splits[chrom].append([sa_chr, chrom, qname, ssp, bool(rev), sp, bool(sam), gs, ge, rs, re_])
        elif sa_chr == chrom and ssp < sp:
            splits[chrom].append([chrom, sa_chr, qname, ssp, bool(rev), sp, bool(sam), gs, ge, rs, re_])
        else:
            splits[chrom].append([chrom, sa_chr, qname, sp, bool(rev), ssp, bool(sam), rs, re_, gs, ge])


_SEL_POOL = None            # pinned host buffers the scan loop's selected reads arrive in (three rotating sets: see _device_scan)


def _device_scan(batch, contig_ok, min_q, max_ins, min_anchor_len, min_clip_len, ctx=None, slot=None, launched=None):
    """slot = 0 / 1 / 2: the three result arrays are views of PINNED buffers of that rotating set (the device-to-host copies are DMA
    transfers, not staged through the runtime's bounce buffers) and stay valid until the set is used again — the scan loop, whose row
    thread is one batch behind, passes batch number mod 3.  None: fresh numpy arrays.  launched(): called when every kernel that reads
    the batch has been enqueued and before the results are waited for (the scan loop starts the next span's inflate there)."""
    global _SEL_POOL
    ctx = ctx or _native.default_context()
    ok = numpy.ascontiguousarray(contig_ok, dtype=numpy.uint8)
    table = (ctypes.c_void_p * 14)(*[batch.dev[k] or None for k in _FIELD_ORDER])
    n_sel, raw_bytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(ctx.lib.tdt_signal_scan(ctx.handle, table, len(batch), _native.ptr(ok), len(ok), int(min_q), int(max_ins), int(min_anchor_len),
                                          int(min_clip_len), ctypes.byref(n_sel), ctypes.byref(raw_bytes)))
    if launched is not None:
        launched()
    if slot is None:
        meta = numpy.empty(n_sel.value, dtype=_META)
        raw_end = numpy.empty(n_sel.value, dtype=numpy.uint32)
        raw = numpy.empty(raw_bytes.value, dtype=numpy.uint8)
    else:
        if _SEL_POOL is None:
            from .hostutil import PinnedPool
            _SEL_POOL = PinnedPool()
        meta = _SEL_POOL.take("meta%d" % slot, n_sel.value, _META)
        raw_end = _SEL_POOL.take("end%d" % slot, n_sel.value, numpy.uint32)
        raw = _SEL_POOL.take("raw%d" % slot, raw_bytes.value, numpy.uint8)
    if n_sel.value:
        _native.check(ctx.lib.tdt_signal_scan_result(ctx.handle, _native.ptr(meta), _native.ptr(raw_end), _native.ptr(raw)))
    return SelectedReads(ctx, meta, raw_end, raw)


def select_discordant(batch, contig_ok, min_q, max_ins, ctx=None):
    """indices (ascending) of the reads of a decoded batch that are discordant-pair signals"""
    ctx = ctx or _native.default_context()
    n = len(batch)
    ok = numpy.ascontiguousarray(contig_ok, dtype=numpy.uint8)
    if isinstance(batch, DeviceBatch):                      # fields already in HBM: predicate + compaction without a re-upload
        import torch
        dev = torch.device("cuda", ctx.device)
        d_ok = torch.from_numpy(ok).to(dev)
        d_out = torch.empty(n, dtype=torch.int32, device=dev)
        d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        d = batch.dev
        _native.check(ctx.lib.tdt_signal_select_device(ctx.handle, d["flag"], d["mapq"], d["tid"], d["mate_tid"], d["tlen"], n,
                                                       d_ok.data_ptr(), len(ok), int(min_q), int(max_ins), d_out.data_ptr(), d_cnt.data_ptr()))
        ctx.sync()
        k = int(d_cnt.item())
        return d_out[:k].cpu().numpy().view(numpy.uint32)
    out = numpy.empty(n, dtype=numpy.uint32)
    cnt = ctypes.c_size_t(0)
    _native.check(ctx.lib.tdt_signal_select(ctx.handle, _native.ptr(batch.flag), _native.ptr(batch.mapq), _native.ptr(batch.tid),
                                            _native.ptr(batch.mate_tid), _native.ptr(batch.tlen), n, _native.ptr(ok), len(ok),
                                            int(min_q), int(max_ins), _native.ptr(out), ctypes.byref(cnt)))
    return out[:cnt.value]


def _scan(bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, bin_size=50, shard=None, reduce_bins=None):
    """One pass over the BAM -> (header, contigs processed, coverage dict, discordant rows, split rows, clip entries, tables).

    Device ingest (the default): ``tables`` is a :class:`sigtab.SignalTables` holding every row and clip entry, merged; the three row
    containers are None.  Host ingest (``TIDDIT_HOST_INGEST=1``): ``tables`` is None and the per-contig Python lists are filled as
    ``worker`` returns them (:228) — discordant rows, split rows, and clip entries ``[header, sequence + "\n"]``.

    shard = (rank, world): only the records that start in this rank's byte range of the file (bamio.DeviceBamReader); the
    seam offsets are left in ``LAST_SEAM`` for dist.check_seams.  reduce_bins(hist) -> float64 array of ALL the histogram's
    bins (the sharded caller all-reduces them there); default: this process's own bins."""
    max_ins = int(max_ins)         # the reference's `int max_ins` argument truncates a float percentile (:147,:230; probed with Cython 3.2)
    from . import bamio
    carry = None
    if os.environ.get("TIDDIT_HOST_INGEST") != "1":
        # the statistics pass of this process (or, on the other ranks of an N-rank job, the pre-ingest that ran beside it) left the head
        # of this very share of the file in HBM; a carry for anything else is dropped there
        carry = bamio.take_carry(bam_file_name, bin_size, shard)
    else:
        bamio.set_carry(None)                                    # (a carry nobody will consume: its batches, reader and histogram go now)
    if carry is not None:
        reader = carry.reader
    elif shard is None:
        reader = open_bam(bam_file_name)
    else:
        reader = DeviceBamReader(bam_file_name, shard=shard, chunk=int(os.environ.get("TIDDIT_INGEST_CHUNK", str(448 << 20))))
    header = reader.header
    names, lengths = reader.references, reader.lengths
    big = numpy.array([ln >= min_contig for ln in lengths], dtype=bool)
    if carry is not None:
        hist = carry.hist                        # (the retained batches' coverage records were written for it)
        hist.reset()
    else:
        hist = tiddit_coverage.CoverageHistogram([(n, l) for n, l in zip(names, lengths)], bin_size)
        if hasattr(reader, "bin_for"):
            reader.bin_for(hist)                 # the ingest kernel writes the coverage records for this bin size
    T = SCAN_SECONDS
    T.clear()
    T.update({"ingest (inflate + decode, device)": 0.0, "coverage push": 0.0})
    tables = data = splits = clips = None
    if isinstance(reader, DeviceBamReader):
        from .sigtab import SignalTables
        tables = SignalTables(names, lengths, min_contig)
        T.update({"predicates + gather of the selected reads (device)": 0.0, "signal tables (native: clip entries, rows, merge; beside the ingest)": 0.0})
    else:
        data = {n: [] for n in names}
        splits = {n: [] for n in names}
        clips = {n: [] for n in names}
        T.update({"field copies + predicates (host)": 0.0, "clip rows": 0.0, "split rows": 0.0, "discordant select + rows": 0.0})

    def rows_of(sel):
        """the selected reads of one batch into the native tables (runs on the worker thread while the device ingests the next batch:
        the main thread waits inside the library, and so does this one — neither holds the GIL)"""
        t4 = time.time()
        tables.add(sel.meta, sel.raw_end, sel.raw, min_q,
                   literal=lambda k: SA_analysis(_ReadProxy(sel, k), min_q, "SA", names[int(sel.tid[k])]))     # (an unusual SA tag: the literal code, and its exceptions)
        T["signal tables (native: clip entries, rows, merge; beside the ingest)"] += time.time() - t4

    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    pending = None
    n_batches = 0
    t0 = time.time()

    def all_batches():
        if carry is None:
            yield from reader.batches()
            return
        for b in carry.batches:                  # already in HBM
            yield b
            b.release()
        carry.batches = []
        yield from carry.iterator                # the rest of the file, from where the statistics pass stopped

    try:
        for b in all_batches():
            t1 = time.time()
            T["ingest (inflate + decode, device)"] += t1 - t0
            if isinstance(b, DeviceBatch):                      # the decoded arrays are already in HBM
                hist.push_device_batch(b, min_q, big)           # coverage (filter on the device, :171-182)
                t3 = time.time()
                T["coverage push"] += t3 - t1
                # the per-read chain of worker (:171-221) on the device; only the selected reads come back (fields + raw records)
                # (once the scan's kernels are enqueued nothing will read the batch's raw bytes again: the next span's inflate is started
                #  behind them — DeviceBamReader.ahead() — and runs while this thread waits for the selected reads and hands them on)
                ahead = getattr(reader, "ahead", None)
                if os.environ.get("TIDDIT_SCAN_RESULT_PINNED", "1") != "0":
                    sel = _device_scan(b, big, min_q, max_ins, min_anchor_len, min_clip_len, slot=n_batches % 3, launched=ahead)
                else:                                           # (as before round 4's last step: pageable arrays, the per-record copies made at once)
                    sel = _device_scan(b, big, min_q, max_ins, min_anchor_len, min_clip_len)
                    sel.raw_bytes, sel.sa_off
                n_batches += 1
                T["predicates + gather of the selected reads (device)"] += time.time() - t3
                if pending is not None:
                    pending.result()                            # batches enter the tables in file order, one batch behind the device
                pending = pool.submit(rows_of, sel)
                t0 = time.time()
                continue
            tid = b.tid
            flag = b.flag.astype(numpy.int32)
            placed = tid >= 0
            ok_contig = numpy.zeros(len(tid), dtype=bool)
            ok_contig[placed] = big[tid[placed]]
            t2 = time.time()
            # coverage: runs of equal tid go to the device as they are (filter on device, :171-182)
            edges = numpy.flatnonzero(numpy.diff(tid)) + 1
            for lo, hi in zip(numpy.concatenate([[0], edges]), numpy.concatenate([edges, [len(tid)]])):
                t = int(tid[lo])
                if t >= 0 and big[t]:
                    hist.push(t, b.pos[lo:hi], b.end[lo:hi], b.mapq[lo:hi], b.flag[lo:hi], min_q)
            t3 = time.time()
            T["coverage push"] += t3 - t2
            primary = ok_contig & ((flag & 0x404) == 0) & ((flag & 0x900) == 0) & (b.mapq >= min_q)   # :171,:184,:188
            same_chr = b.mate_tid == tid
            abs_isize = numpy.abs(b.tlen.astype(numpy.int64))
            # clipped reads for local assembly (:190-197)
            f_op, f_len = b.cigar_first & 0xf, b.cigar_first >> 4
            l_op, l_len = b.cigar_last & 0xf, b.cigar_last >> 4
            has_cigar = b.cigar_first != 0xffffffff
            left = (f_op == 4) & (f_len > min_clip_len) & (l_op == 0) & (l_len > min_anchor_len)
            right = (l_op == 4) & (l_len > min_clip_len) & (f_op == 0) & (f_len > min_anchor_len)
            clip_idx = numpy.flatnonzero(primary & (abs_isize < max_ins) & same_chr & has_cigar & (left | right))
            split_idx = numpy.flatnonzero(primary & (b.sa_off >= 0))
            t4 = time.time()
            T["field copies + predicates (host)"] += (t2 - t1) + (t4 - t3)
            for i in clip_idx:
                rec = b.record(i)
                chrom = names[tid[i]]
                clips[chrom].append([">{}|{}|{}\n".format(rec.query_name, chrom, int(b.pos[i]) + 1), rec.query_sequence + "\n"])
            t5 = time.time()
            T["clip rows"] += t5 - t4
            # split reads (:199-202)
            for i in split_idx:
                chrom = names[tid[i]]
                split = SA_analysis(_ReadProxy(b, i), min_q, "SA", chrom)
                if split:
                    splits[chrom].append(split)
            t6 = time.time()
            T["split rows"] += t6 - t5
            # discordant pairs (:204-221): predicate + order-preserving compaction on the device
            for i in select_discordant(b, big, min_q, max_ins):
                chrom, mate = names[tid[i]], names[b.mate_tid[i]]
                chrA, chrB = (mate, chrom) if mate < chrom else (chrom, mate)
                data[chrom].append([chrA, chrB, b.record(i).query_name, int(b.pos[i]) + 1, int(b.end[i]) + 1, bool(flag[i] & 0x10), chrom])
            t0 = time.time()
            T["discordant select + rows"] += t0 - t6
        if pending is not None:
            pending.result()
    except BaseException:
        if tables is not None:
            pool.shutdown(wait=True)
            tables.close()
        raise
    finally:
        pool.shutdown(wait=True)                              # (also on an error: no row thread outlives the scan)
    if shard is not None:
        LAST_SEAM.update(first_off=reader.first_off, next_off=reader.next_off, empty=reader.first_off is None)
    if getattr(reader, "reader_seconds", None):
        READER_SECONDS.clear()
        READER_SECONDS.update(reader.reader_seconds)             # (the statistics pass's share included when its reader was carried over)
        if getattr(reader, "timings", None):                     # TIDDIT_INGEST_TIMING=1: the pushes' own stage times, summed
            for k_ in ("h2d_ms", "inflate_crc_ms", "find_records_ms", "chain_check_ms", "decode_ms", "push_wall_ms", "block_table_ms"):
                READER_SECONDS["push: " + k_[:-3] + " (s)"] = sum(t_[k_] for t_ in reader.timings) * 1e-3
            READER_SECONDS["pushes"] = len(reader.timings)
    reader.close()
    chromosomes = [n for n, ok in zip(names, big) if ok]
    if reduce_bins is None and len(chromosomes) <= 64:
        coverage = {n: hist.finish(n) for n in chromosomes}
    else:
        # (a header of hundreds of kept contigs — GRCh38's alt contigs — or the N-rank job's reduced bins: every contig's bins in one piece)
        allbins = reduce_bins(hist) if reduce_bins is not None else hist.finish_all()
        coverage = {}
        for i, n in enumerate(names):
            if big[i]:
                o = hist.offset(i)
                coverage[n] = allbins[o:o + hist.nbins(i)[0]].copy()
    hist.close()
    return header, chromosomes, coverage, data, splits, clips, tables


def scan_signals(bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, bin_size=50, shard=None, reduce_bins=None):
    """:func:`_scan` with the rows as Python lists whichever way the file was read: -> (header, contigs processed, coverage dict,
    per-contig discordant rows, split rows, clip FASTA entries) — what the reference's ``worker`` returns per contig (:228).  A clip
    entry is a pair whose joined content is FASTA text: ``[header, sequence + "\n"]`` strings, one per read, with the host ingest;
    ``[bytes, ""]``, the contig's whole text, with the device ingest (``_clip_bytes(entry)`` gives the bytes of either)."""
    header, chromosomes, coverage, data, splits, clips, tables = _scan(bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len,
                                                                       bin_size, shard, reduce_bins)
    if tables is not None:
        data, splits = tables.rows()
        clips = {n: ([[tables.clips(t), ""]] if tables.clips(t) else []) for t, n in enumerate(tables.names)}
        tables.close()
    return header, chromosomes, coverage, data, splits, clips


_SCAN_CACHE = {}
STAGE_SECONDS = {}          # wall seconds of the last main(), stage by stage
SCAN_SECONDS = {}           # ... and of the last scan pass, by what the host waited for
LAST_SEAM = {}              # seam offsets of the last sharded scan pass (dist.check_seams)
AFTER_SCAN = []             # callables main() invokes once the file has been scanned, before the tables are written (host-only work from there on)
READER_SECONDS = {}         # the reader thread of the last scan: seconds reading, scanning BGZF headers, building tables, waiting — and the consumer's waits
WRITTEN_TABLES = {}         # (discordants path, splits path) -> stamps + the native tables the last main() wrote them from (tiddit_cluster takes them over)


def worker(chromosome, bam_file_name, ref, prefix, min_q, max_ins, sample_id, bin_size, skip_index, min_anchor_len, min_clip_len):
    """One contig's share of the scan with the reference's signature and return value (tiddit_signal.pyx:147-228):
    ``(chromosome, discordant rows, split rows, float64 coverage bins, path of the clipped-read FASTA)``.  The reference runs
    one indexed pysam pass per contig in a joblib worker; here ONE device pass over the file serves every contig (cached per
    file and parameter set), so calling ``worker`` contig by contig costs one scan in total."""
    max_ins = int(max_ins)
    key = (os.path.abspath(bam_file_name), os.path.getmtime(bam_file_name), min_q, max_ins, bin_size, min_anchor_len, min_clip_len)
    if key not in _SCAN_CACHE:
        _SCAN_CACHE.clear()
        _SCAN_CACHE[key] = scan_signals(bam_file_name, min_q, max_ins, 0, min_anchor_len, min_clip_len, bin_size)
    header, chromosomes, coverage, data, splits, clips = _SCAN_CACHE[key]
    print("Collecting signals on contig: {}".format(chromosome))
    os.makedirs("{}_tiddit/clips".format(prefix), exist_ok=True)
    path = "{}_tiddit/clips/{}.fa".format(prefix, chromosome)
    with open(path, "wb") as f:
        for clip in clips[chromosome]:
            f.write(_clip_bytes(clip))
    return (chromosome, data[chromosome], splits[chromosome], coverage[chromosome], path)


def _merge_and_write(header, chromosomes, res_data, res_splits, res_clips, prefix, sample_id):
    """the merge loop and the three writers of tiddit_signal.main (:246-332) over per-contig row lists in file order — the literal
    Python form (host ingest; the device ingest's rows never leave the native tables: :func:`_write_tables`)"""
    all_contigs = [c["SN"] for c in header["SQ"]]
    data = {a: {b: {} for b in all_contigs} for a in chromosomes}        # :246-256
    splits = {a: {b: {} for b in all_contigs} for a in chromosomes}
    os.makedirs("{}_tiddit/clips".format(prefix), exist_ok=True)
    with open("{}_tiddit/clips_{}.fa".format(prefix, sample_id), "wb") as all_clips:      # (written last in the reference; same bytes)
        for chrom in chromosomes:                                            # results in contig order (:262-284)
            print("Collecting signals on contig: {}".format(chrom))
            for signal in res_data[chrom]:
                if signal[0] not in data:
                    continue
                data[signal[0]][signal[1]].setdefault(signal[2], []).append(signal[3:])
            for signal in res_splits[chrom]:
                if signal[0] not in splits:
                    continue
                splits[signal[0]][signal[1]].setdefault(signal[2], [])
                splits[signal[0]][signal[1]][signal[2]] += signal[3:]
            path = "{}_tiddit/clips/{}.fa".format(prefix, chrom)
            with open(path, "wb") as f:
                for clip in res_clips[chrom]:
                    text = _clip_bytes(clip)
                    f.write(text)
                    all_clips.write(text)                # clips_{sample}.fa is the per-contig files one after the other (:328-332)
    print("Writing signals to file")
    disc_path, split_path = "{}_tiddit/discordants_{}.tab".format(prefix, sample_id), "{}_tiddit/splits_{}.tab".format(prefix, sample_id)
    with open(disc_path, "w") as f:      # :298-318
        for chrA in data:
            for chrB in data[chrA]:
                for fragment, reads in data[chrA][chrB].items():
                    if len(reads) < 2:
                        continue
                    first, second = reads[0], reads[1]
                    if chrA == chrB:
                        if second[-1] < first[-1]:      # QUIRK (:307): compares the two read_chr strings, always equal
                            first, second = second, first
                    elif first[-1] != chrA:
                        first, second = second, first
                    out = first[0:-1] + second[0:-1]
                    f.write("{}\t{}\t{}\t{}\n".format(fragment, chrA, chrB, "\t".join(map(str, out))))
    with open(split_path, "w") as f:           # :320-326
        for chrA in splits:
            for chrB in splits[chrA]:
                for fragment, fields in splits[chrA][chrB].items():
                    f.write("{}\t{}\t{}\t{}\n".format(fragment, chrA, chrB, "\t".join(map(str, fields))))
    _forget_tables()


def _write_tables(scanned, merged, chromosomes, prefix, sample_id, group=None, owner=None):
    """The three writers of tiddit_signal.main (:298-332) from the native tables: clips/{contig}.fa and their concatenation
    clips_{sample}.fa, discordants_{sample}.tab, splits_{sample}.tab.  Every file is a sequence of per-contig blocks in header order
    (the rows of one chrA; one contig's clip entries), so the writers never hold a file in Python: the library writes each block at its
    offset (``tdt_sigtab_pwrite``).  On N ranks — `scanned` holds the clip entries of this rank's share of the file, `merged` the rows
    of the chrA this rank owns — the block sizes of every rank meet in one all-gather, rank 0 creates the files at their final size and
    every rank places its own blocks: nothing but the sizes travels.  (One node, one output directory: BASELINE configs[4].)
    `merged` stays alive for tiddit_cluster.main of this process (:func:`written_tables`)."""
    names = merged.names
    n = len(names)
    kept = [i for i, ln in enumerate(merged.lengths) if ln >= merged.min_contig]                 # main()'s `chromosomes`, as contig ids
    mine = numpy.stack([merged.sizes(0), merged.sizes(1), scanned.sizes(2)])                     # [what][contig]
    rank, world = 0, 1
    if owner is not None:
        import torch.distributed as dist
        from . import dist as tdist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        sizes = tdist.allgather_i64(mine.reshape(-1), group).reshape(world, 3, n)
    else:
        sizes = mine.reshape(1, 3, n)
    total = sizes.sum(axis=0)                                                                    # [what][contig]
    before = numpy.cumsum(sizes, axis=0) - sizes                                                 # bytes of the earlier ranks, per block
    d_path, s_path = "{}_tiddit/discordants_{}.tab".format(prefix, sample_id), "{}_tiddit/splits_{}.tab".format(prefix, sample_id)
    all_path = "{}_tiddit/clips_{}.fa".format(prefix, sample_id)
    clip_path = lambda t: "{}_tiddit/clips/{}.fa".format(prefix, names[t])
    if rank == 0:
        os.makedirs("{}_tiddit/clips".format(prefix), exist_ok=True)
        for c in chromosomes:
            print("Collecting signals on contig: {}".format(c))
        print("Writing signals to file")
        for path, size in [(d_path, int(total[0][kept].sum())), (s_path, int(total[1][kept].sum())), (all_path, int(total[2][kept].sum()))] + \
                          [(clip_path(t), int(total[2][t])) for t in kept]:
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
            os.ftruncate(fd, size)
            os.close(fd)
    if world > 1:
        dist.barrier(group)
    base = numpy.zeros((3, n), dtype=numpy.int64)                                                # where a contig's block starts in the three big files
    for w in range(3):
        base[w][kept] = numpy.cumsum(total[w][kept]) - total[w][kept]

    def place_blocks():
        # every block is independent of every other (its own bytes, its own offset; tdt_sigtab_pwrite only reads the tables and drops the
        # GIL), so a few threads place them side by side: the clip FASTA of a 3-Gb sample is 0.3 GB, written twice — 0.13 s on one thread,
        # longer than the ploidy table and the clustering it runs beside
        fds, tasks = {}, []

        def whole(tab, w, t, path, off):
            fd = fds.get(path)
            if fd is None:
                fd = fds[path] = os.open(path, os.O_WRONLY)
            tasks.append((tab, w, t, fd, off, None))

        try:
            for w, path in ((0, d_path), (1, s_path)):
                for t in kept:
                    if mine[w][t]:
                        whole(merged, w, t, path, int(base[w][t] + before[rank][w][t]))
            for t in kept:
                if mine[2][t]:
                    whole(scanned, 2, t, all_path, int(base[2][t] + before[rank][2][t]))
                    tasks.append((scanned, 2, t, None, int(before[rank][2][t]), clip_path(t)))

            def place(task):
                tab, w, t, fd, off, own = task
                if own is None:
                    tab.pwrite(w, t, fd, off)
                    return
                fd = os.open(own, os.O_WRONLY)
                try:
                    tab.pwrite(w, t, fd, off)
                finally:
                    os.close(fd)

            workers = min(len(tasks), int(os.environ.get("TIDDIT_WRITE_THREADS", "4")))
            if workers > 1:
                with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as ex:
                    list(ex.map(place, tasks))                  # (the first error of any block is raised here)
            else:
                for task in tasks:
                    place(task)
        finally:
            for fd in fds.values():
                os.close(fd)

    key = (os.path.abspath(d_path), os.path.abspath(s_path))
    _forget_tables()
    if BACKGROUND_WRITES and world == 1:
        # The one-process `tiddit --sv` sets this: the text is formatted (merged.sizes above), the files exist at their final size, and
        # the blocks are placed by a thread (pwrite in the library, no GIL) while the job goes on to the ploidy table and the clustering
        # — which takes the TABLES over, not the files.  finish_writes() waits for it; until then the entry is marked pending.
        job = {"seconds": 0.0}

        def run():
            t0 = time.time()
            try:
                place_blocks()
                if scanned is not merged:
                    scanned.close()
            except BaseException as e:                          # re-raised by finish_writes()
                job["error"] = e
            job["seconds"] = time.time() - t0

        job["thread"] = threading.Thread(target=run, name="tiddit-signal-writer")
        WRITTEN_TABLES[key] = (None, None, merged, owner, job)
        job["thread"].start()
        return
    place_blocks()
    if world > 1:
        dist.barrier(group)                                      # the files are complete when any rank returns
    WRITTEN_TABLES[key] = (_file_stamp(d_path), _file_stamp(s_path), merged, owner, None)
    if scanned is not merged:
        scanned.close()


BACKGROUND_WRITES = False   # see _write_tables; a library caller of main() gets complete files on return
WRITE_SECONDS = {}          # what the last finish_writes() found: seconds the writer thread ran, seconds the caller waited for it


def finish_writes():
    """Wait for the writer thread of a main() that ran with BACKGROUND_WRITES (a no-op otherwise): the three files are complete on
    return, and the tables' hand-over entry carries their stamps like any other."""
    WRITE_SECONDS.clear()
    for key, ent in list(WRITTEN_TABLES.items()):
        job = ent[4]
        if job is None:
            continue
        t0 = time.time()
        job["thread"].join()
        WRITE_SECONDS["writer thread"] = job["seconds"]
        WRITE_SECONDS["waited for it"] = time.time() - t0
        if "error" in job:
            ent[2].close()
            del WRITTEN_TABLES[key]
            raise job["error"]
        WRITTEN_TABLES[key] = (_file_stamp(key[0]), _file_stamp(key[1]), ent[2], ent[3], None)


def _forget_tables():
    for ent in WRITTEN_TABLES.values():
        if ent[4] is not None:
            ent[4]["thread"].join()
        ent[2].close()
    WRITTEN_TABLES.clear()


def _file_stamp(path):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns, st.st_ino)


def written_tables(disc_path, split_path):
    """the native signal tables (sigtab.SignalTables) of the last `main` of THIS process if the two files on disk are still the ones
    it wrote (size, mtime, inode) — else None, and the caller parses the text."""
    ent = WRITTEN_TABLES.get((os.path.abspath(disc_path), os.path.abspath(split_path)))
    if ent is None:
        return None
    if ent[4] is not None:
        return ent[2]               # (this process is placing the blocks of exactly these tables right now: BACKGROUND_WRITES)
    try:
        if _file_stamp(disc_path) != ent[0] or _file_stamp(split_path) != ent[1]:
            return None
    except OSError:
        return None
    return ent[2]


def table_owners(disc_path, split_path):
    """the owner rank of every contig's rows when the tables :func:`written_tables` returns are one rank's share of an N-rank job
    (else None)"""
    ent = WRITTEN_TABLES.get((os.path.abspath(disc_path), os.path.abspath(split_path)))
    return None if ent is None else ent[3]


def _main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len):
    t = time.time()
    header, chromosomes, coverage_data, res_data, res_splits, res_clips, tables = _scan(
        bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, 50)
    STAGE_SECONDS.clear()
    STAGE_SECONDS["scan (ingest, coverage, predicates, signal tables)"] = time.time() - t
    STAGE_SECONDS.update({"  " + k: v for k, v in SCAN_SECONDS.items()})
    for hook in list(AFTER_SCAN):
        hook()
    t1 = time.time()
    if tables is not None:
        _write_tables(tables, tables, chromosomes, prefix, sample_id)
    else:
        _merge_and_write(header, chromosomes, res_data, res_splits, res_clips, prefix, sample_id)
    print("total", time.time() - t)
    STAGE_SECONDS["write .tab / clips" if tables is not None else "merge + write .tab / clips"] = time.time() - t1
    return coverage_data


def _main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group):
    """tiddit_signal.main with one process per GPU on ONE file (BASELINE configs[4]).  The reference fans out one worker per contig and
    merges their rows in contig order on one core (:259-284).  Here rank r scans the records that start in its 1/N of the file's bytes
    (BGZF blocks are independent; the seams are checked, dist.check_seams) into its own native tables; the 50-bp bins meet in ONE exact
    all-reduce; and the rows travel ONCE, in one all-to-all, to the owner rank of their chrA (dist.contig_owners: contigs bin-packed by
    length).  The owner appends what it receives in RANK order — the file is coordinate sorted, so that is the file order of the
    single-process scan, and a fragment whose two reads were scanned by different ranks is paired exactly as in :262-284 — formats its
    chrA's rows and places them in the .tab files itself (:func:`_write_tables`); the clip FASTA blocks are placed by the ranks that
    scanned them.  No rank gathers rows, and no rank parses text later: tiddit_cluster.main_sharded takes the owner's tables over."""
    import torch
    import torch.distributed as dist
    from . import dist as tdist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t = time.time()

    def reduce_bins(hist):
        ctx = hist.ctx
        dev = torch.device("cuda", ctx.device)
        bins = torch.empty(hist.total_bins(), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)          # torch's allocator work runs on torch's stream, the library on its own: order them
        hist.finish_all_device(bins.data_ptr())
        ctx.sync()
        tdist.check_seams(LAST_SEAM["first_off"], LAST_SEAM["next_off"], LAST_SEAM["empty"], group)
        if dist.get_backend(group) != "nccl":
            bins = tdist.allreduce_bins(bins.cpu(), group)
        else:
            tdist.allreduce_bins(bins, group)
        return bins.cpu().numpy()

    header, chromosomes, coverage_data, _, _, _, scanned = _scan(
        bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, 50, shard=(rank, world), reduce_bins=reduce_bins)
    STAGE_SECONDS.clear()
    STAGE_SECONDS["scan (ingest, coverage, predicates, signal tables; this rank's shard)"] = time.time() - t
    STAGE_SECONDS.update({"  " + k: v for k, v in SCAN_SECONDS.items()})
    for hook in list(AFTER_SCAN):
        hook()
    share_and_write(scanned, chromosomes, prefix, sample_id, group)
    return coverage_data


def share_and_write(scanned, chromosomes, prefix, sample_id, group=None):
    """the N-rank job behind the scan: this rank's rows to the owner ranks of their chrA (ONE all-to-all), the owner's merge in rank
    order, and the output files from every rank's own blocks (:func:`_write_tables`)"""
    import torch.distributed as dist
    from . import dist as tdist
    from .sigtab import SignalTables
    world = dist.get_world_size(group)
    t1 = time.time()
    owner = tdist.contig_owners(scanned.lengths, [ln >= scanned.min_contig for ln in scanned.lengths], world)
    got = tdist.alltoall_bytes([scanned.export_rows(owner, r) for r in range(world)], group)
    merged = SignalTables(scanned.names, scanned.lengths, scanned.min_contig)
    for blob in got:                                             # rank order = file order inside every contig
        merged.import_rows(blob)
    STAGE_SECONDS["rows to their owner ranks (all-to-all) + merge"] = time.time() - t1
    t1 = time.time()
    _write_tables(scanned, merged, chromosomes, prefix, sample_id, group=group, owner=owner)
    STAGE_SECONDS["write .tab / clips (every rank its own blocks)"] = time.time() - t1
    return merged, owner


def main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group=None):
    """:func:`main` with one process per GPU (torch.distributed initialised; nccl = RCCL, or gloo) — see :func:`_main_sharded`"""
    with quiet_gc():
        return _main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group)


def main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len):
    """``tiddit_signal.main`` (tiddit_signal.pyx:230-334): signals of every contig -> discordants_/splits_ .tab, clips_ .fa; returns the
    50-bp coverage dictionary."""
    with quiet_gc():
        return _main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len)
