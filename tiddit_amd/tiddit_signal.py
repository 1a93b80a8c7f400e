"""Drop-in for ``tiddit.tiddit_signal`` (tiddit_signal.pyx) on the MI355X.

``main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index,
min_anchor_len, min_clip_len) -> coverage_data`` (:230-334) with the same side-effect files
(``{prefix}_tiddit/discordants_{sample}.tab``, ``splits_{sample}.tab``, ``clips_{sample}.fa``,
``clips/{contig}.fa``), plus ``SA_analysis`` / ``find_SA_query_range`` (:11-145).

Redesign: the reference forks one joblib worker per contig and calls Python once per read
(``worker`` :147-228).  Here ONE process streams the BAM through the C record decoder
(``bamio.BamReader``); the packed start/end/mapq/flag arrays of every batch go to the device coverage
histogram (bin 50, same read filter, :171-182), and the signal predicates (:184-221) are evaluated on
whole arrays — only the ~1 % of reads that are discordant, split or clipped are touched one by one.
``threads`` and ``skip_index`` are accepted for signature compatibility and ignored (no index needed).
"""
import concurrent.futures
import itertools
import os
import re
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
        self.raw_bytes = raw.tobytes()                  # the selected records are ~1.5 % of the batch: RecordView slices this copy
        self.tid, self.pos, self.end, self.flag, self.mate_tid, self.action = (meta[k] for k in ("tid", "pos", "end", "flag", "mate_tid", "action"))
        self.rec_off = numpy.concatenate([[0], raw_end[:-1]]).astype(numpy.uint64) if len(raw_end) else numpy.zeros(0, dtype=numpy.uint64)
        self.sa_off = numpy.where(meta["sa_rel"] >= 0, self.rec_off.astype(numpy.int64) + meta["sa_rel"], -1)

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


def _device_scan(batch, contig_ok, min_q, max_ins, min_anchor_len, min_clip_len, ctx=None):
    ctx = ctx or _native.default_context()
    ok = numpy.ascontiguousarray(contig_ok, dtype=numpy.uint8)
    table = (ctypes.c_void_p * 14)(*[batch.dev[k] or None for k in _FIELD_ORDER])
    n_sel, raw_bytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(ctx.lib.tdt_signal_scan(ctx.handle, table, len(batch), _native.ptr(ok), len(ok), int(min_q), int(max_ins), int(min_anchor_len),
                                          int(min_clip_len), ctypes.byref(n_sel), ctypes.byref(raw_bytes)))
    meta = numpy.empty(n_sel.value, dtype=_META)
    raw_end = numpy.empty(n_sel.value, dtype=numpy.uint32)
    raw = numpy.empty(raw_bytes.value, dtype=numpy.uint8)
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


class EarlyTables:
    """The (chrA, chrB, fragment) dictionaries of tiddit_signal.main (:246-284), filled WHILE the file is scanned.

    main() walks the contigs in header order and, inside a contig, its rows in file order.  On a coordinate-sorted file whose contig
    order is the header's that IS the order in which ``scan_signals`` produces rows, so a batch's rows can be merged as soon as they
    exist — on the worker thread, behind the device ingest of the next batch — and main() finds the dictionaries ready.  A row out
    of contig order switches the early merge off (``ok`` False); main() then merges the per-contig lists as before.

    ``data[chrA][chrB][fragment]`` = the fragment's reads (each ``row[3:]``) as in the reference, and — once the second read has
    arrived, i.e. the row of :298-318 is complete — two more entries: ``(fragment, chrA, chrB, fields as written)`` and the text line.
    ``splits[chrA][chrB][fragment]`` = the concatenated fields (:282); ``slines[..]`` = the line as main() writes it."""

    def __init__(self, all_contigs, kept_contigs, enabled=True):
        self.ok = bool(enabled)
        self.rank_of = {c: i for i, c in enumerate(kept_contigs)}          # chromosomes of main(): the contigs of at least min_contig, in header order
        self.last = {"d": -1, "s": -1}
        self.data = {a: {b: {} for b in all_contigs} for a in self.rank_of}
        self.splits = {a: {b: {} for b in all_contigs} for a in self.rank_of}
        self.slines = {a: {b: {} for b in all_contigs} for a in self.rank_of}

    def add(self, chrom, rows, which):
        """rows of contig `chrom` (file order), which = "d" (discordant rows of worker, :214-221) or "s" (split rows)"""
        r = self.rank_of.get(chrom)
        if r is None or not self.ok:
            return
        if r < self.last[which]:
            self.ok = False
            return
        self.last[which] = r
        if which == "d":
            tab = self.data
            for signal in rows:
                chrA = signal[0]
                a = tab.get(chrA)
                if a is not None:
                    chrB = signal[1]
                    reads = a[chrB].setdefault(signal[2], [])
                    reads.append(signal[3:])
                    if len(reads) == 2:
                        first, second = reads
                        if chrA == chrB:
                            if second[-1] < first[-1]:           # QUIRK (:307): compares the two read_chr strings, always equal
                                first, second = second, first
                        elif first[-1] != chrA:
                            first, second = second, first
                        out = first[0:-1] + second[0:-1]
                        reads.append((signal[2], chrA, chrB, out))
                        reads.append("{}\t{}\t{}\t{}\n".format(signal[2], chrA, chrB, "\t".join(map(str, out))))
        else:
            tab, lines = self.splits, self.slines
            for signal in rows:
                chrA = signal[0]
                a = tab.get(chrA)
                if a is not None:
                    f = a[signal[1]].setdefault(signal[2], [])
                    f += signal[3:]
                    lines[chrA][signal[1]][signal[2]] = "{}\t{}\t{}\t{}\n".format(signal[2], chrA, signal[1], "\t".join(map(str, f)))


def scan_signals(bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, bin_size=50, shard=None, reduce_bins=None):
    """One pass over the BAM: -> (header, contigs processed, coverage dict, per-contig discordant rows,
    split rows, clip FASTA entries).  The discordant and split rows are what ``worker`` returns (:228); the clip entries
    are pairs whose joined content is the contig's clip FASTA in file order: ``[header, sequence + "\n"]`` strings, one per read,
    with the host ingest; ``[bytes, ""]``, one per contig run of a batch, already formatted by ``tdt_format_clips``, with the
    device ingest.  ``_clip_bytes(entry)`` gives the bytes of either — joining is what every consumer does (:223-226).

    shard = (rank, world): only the records that start in this rank's byte range of the file (bamio.DeviceBamReader); the
    seam offsets are left in ``LAST_SEAM`` for dist.check_seams.  reduce_bins(hist) -> float64 array of ALL the histogram's
    bins (the sharded caller all-reduces them there); default: this process's own bins."""
    max_ins = int(max_ins)         # the reference's `int max_ins` argument truncates a float percentile (:147,:230; probed with Cython 3.2)
    carry = None
    if shard is None and os.environ.get("TIDDIT_HOST_INGEST") != "1":
        from . import bamio
        carry = bamio.take_carry(bam_file_name, bin_size)      # the statistics pass of this process left its sampled batches in HBM
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
    data = {n: [] for n in names}
    splits = {n: [] for n in names}
    clips = {n: [] for n in names}
    T = SCAN_SECONDS
    T.clear()
    T.update({"ingest (inflate + decode, device)": 0.0, "coverage push": 0.0, "field copies + predicates (host)": 0.0, "clip rows": 0.0,
              "split rows": 0.0, "discordant select + rows": 0.0})
    # the per-fragment merge of main() runs while the file is scanned (EarlyTables)
    early = EarlyTables(list(names), [names[t] for t, ok in enumerate(big) if ok], enabled=shard is None)
    merge_early = early.add

    def rows_of(sel):
        """clip / split / discordant rows of one batch's selected reads (host copies only: runs on the worker thread while the device
        ingests the next batch — the main thread waits inside the library without the GIL)"""
        t4 = time.time()
        stid, act = sel.tid, sel.action
        clip_k = numpy.flatnonzero(act & 2)
        if len(clip_k):
            edges = numpy.flatnonzero(numpy.diff(stid[clip_k])) + 1
            for lo, hi in zip(numpy.concatenate([[0], edges]), numpy.concatenate([edges, [len(clip_k)]])):
                chrom = names[stid[clip_k[lo]]]
                clips[chrom].append([sel.clip_fasta(clip_k[lo:hi], chrom), ""])
        t5 = time.time()
        T["clip rows"] += t5 - t4
        which4 = numpy.flatnonzero(act & 4)
        tids4 = stid[which4]
        runs4 = [] if not len(which4) else [int(t) for t in tids4[numpy.concatenate([[0], numpy.flatnonzero(numpy.diff(tids4)) + 1])]]
        before = {t: len(splits[names[t]]) for t in set(runs4)}
        split_rows_native(sel, which4, names, min_q, splits)
        if early.ok:
            if len(set(runs4)) != len(runs4):
                early.ok = False                                 # a contig twice in one batch: not coordinate sorted
            for t in runs4:
                merge_early(names[t], splits[names[t]][before[t]:], "s")
        t6 = time.time()
        T["split rows"] += t6 - t5
        which = numpy.flatnonzero(act & 8)
        rb = sel.raw_bytes
        cols = zip(stid[which].tolist(), sel.mate_tid[which].tolist(), sel.pos[which].tolist(), sel.end[which].tolist(),
                   sel.flag[which].tolist(), sel.rec_off[which].tolist())
        cur_t, cur_rows = None, None
        for t_, m_, p_, e_, f_, o_ in cols:
            chrom, mate = names[t_], names[m_]
            chrA, chrB = (mate, chrom) if mate < chrom else (chrom, mate)
            qname = rb[o_ + 36:o_ + 35 + rb[o_ + 12]].decode()          # block_size, 32 fixed bytes, then l_read_name bytes (NUL included)
            row = [chrA, chrB, qname, p_ + 1, e_ + 1, bool(f_ & 0x10), chrom]
            data[chrom].append(row)
            if t_ != cur_t:
                if cur_rows:
                    merge_early(names[cur_t], cur_rows, "d")
                cur_t, cur_rows = t_, []
            cur_rows.append(row)
        if cur_rows:
            merge_early(names[cur_t], cur_rows, "d")
        T["discordant select + rows"] += time.time() - t6

    pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
    pending = None
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
            if not isinstance(b, DeviceBatch):
                tid = b.tid
                flag = b.flag.astype(numpy.int32)
                placed = tid >= 0
                ok_contig = numpy.zeros(len(tid), dtype=bool)
                ok_contig[placed] = big[tid[placed]]
            t2 = time.time()
            # coverage: runs of equal tid go to the device as they are (filter on device, :171-182)
            if isinstance(b, DeviceBatch):                      # the decoded arrays are already in HBM
                hist.push_device_batch(b, min_q, big)
            else:
                edges = numpy.flatnonzero(numpy.diff(tid)) + 1
                for lo, hi in zip(numpy.concatenate([[0], edges]), numpy.concatenate([edges, [len(tid)]])):
                    t = int(tid[lo])
                    if t >= 0 and big[t]:
                        hist.push(t, b.pos[lo:hi], b.end[lo:hi], b.mapq[lo:hi], b.flag[lo:hi], min_q)
            t3 = time.time()
            T["coverage push"] += t3 - t2
            if isinstance(b, DeviceBatch):
                # the per-read chain of worker (:171-221) on the device; only the selected reads come back (fields + raw records)
                sel = _device_scan(b, big, min_q, max_ins, min_anchor_len, min_clip_len)
                t4 = time.time()
                T["predicates + gather of the selected reads (device)"] = T.get("predicates + gather of the selected reads (device)", 0.0) + (t4 - t3) + (t2 - t1)
                if pending is not None:
                    pending.result()                                  # rows are built in batch order, one batch behind the device
                pending = pool.submit(rows_of, sel)
                t0 = time.time()
                continue
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
    finally:
        pool.shutdown(wait=True)                              # (also on an error: no row thread outlives the scan)
    if shard is not None:
        LAST_SEAM.update(first_off=reader.first_off, next_off=reader.next_off, empty=reader.first_off is None)
    reader.close()
    chromosomes = [n for n, ok in zip(names, big) if ok]
    if reduce_bins is None:
        coverage = {n: hist.finish(n) for n in chromosomes}
    else:
        allbins = reduce_bins(hist)
        coverage = {}
        for i, n in enumerate(names):
            if big[i]:
                o = hist.offset(i)
                coverage[n] = allbins[o:o + hist.nbins(i)[0]].copy()
    hist.close()
    PREMERGED.clear()
    if early.ok and isinstance(reader, DeviceBamReader):
        PREMERGED["tables"] = (early.data, early.splits, data, splits, early.slines)   # (keyed to the very lists main() merges)
    return header, chromosomes, coverage, data, splits, clips


_SCAN_CACHE = {}
STAGE_SECONDS = {}          # wall seconds of the last main(), stage by stage
SCAN_SECONDS = {}           # ... and of the last scan_signals() pass, by what the host waited for
LAST_SEAM = {}              # seam offsets of the last sharded scan_signals() pass (dist.check_seams)
PREMERGED = {}              # the (chrA, chrB, fragment) dictionaries the last scan_signals() merged while it scanned (see rows_of)
AFTER_SCAN = []             # callables main() invokes once the file has been scanned, before the tables are merged and written (host-only work from there on)
WRITTEN_TABLES = {}         # (discordants path, splits path) -> stamps + the rows the last main() wrote there (tiddit_cluster reads them back)


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
    """the merge loop and the three writers of tiddit_signal.main (:246-332) over per-contig row lists in file order"""
    all_contigs = [c["SN"] for c in header["SQ"]]
    pre = PREMERGED.pop("tables", None)
    if pre is not None and pre[2] is res_data and pre[3] is res_splits and list(pre[0]) == list(chromosomes):
        data, splits = pre[0], pre[1]                                        # merged while the file was scanned (scan_signals)
    else:
        pre = None
        data = {a: {b: {} for b in all_contigs} for a in chromosomes}        # :246-256
        splits = {a: {b: {} for b in all_contigs} for a in chromosomes}
    os.makedirs("{}_tiddit/clips".format(prefix), exist_ok=True)
    clip_fasta = []
    with open("{}_tiddit/clips_{}.fa".format(prefix, sample_id), "wb") as all_clips:      # (written last in the reference; same bytes)
        for chrom in chromosomes:                                            # results in contig order (:262-284)
            print("Collecting signals on contig: {}".format(chrom))
            for signal in (res_data[chrom] if pre is None else ()):
                if signal[0] not in data:
                    continue
                data[signal[0]][signal[1]].setdefault(signal[2], []).append(signal[3:])
            for signal in (res_splits[chrom] if pre is None else ()):
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
            clip_fasta.append(path)
    print("Writing signals to file")

    # the rows as written, per contig pair in file order: tiddit_cluster in the same process takes them from here (no text re-parse).
    # disc_rows: (chrA, chrB, [(fragment, chrA, chrB, fields), ...]); split_rows: (chrA, chrB, {fragment: fields})
    disc_rows, split_rows = [], []
    disc_path, split_path = "{}_tiddit/discordants_{}.tab".format(prefix, sample_id), "{}_tiddit/splits_{}.tab".format(prefix, sample_id)
    with open(disc_path, "w") as f:      # :298-318
        for chrA in data:
            for chrB in data[chrA]:
                if pre is not None:                     # the rows were formatted when their second read arrived: same order, same text
                    frags = data[chrA][chrB]
                    if frags:
                        rows = [reads[2] for reads in frags.values() if len(reads) > 2]
                        f.write("".join([reads[3] for reads in frags.values() if len(reads) > 2]))
                        if rows:
                            disc_rows.append((chrA, chrB, rows))
                    continue
                rows = []
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
                    rows.append((fragment, chrA, chrB, out))
                if rows:
                    disc_rows.append((chrA, chrB, rows))
    with open(split_path, "w") as f:           # :320-326
        for chrA in splits:
            for chrB in splits[chrA]:
                if pre is not None:
                    frags = splits[chrA][chrB]
                    if frags:
                        f.write("".join(pre[4][chrA][chrB].values()))
                        split_rows.append((chrA, chrB, frags))
                    continue
                for fragment, fields in splits[chrA][chrB].items():
                    f.write("{}\t{}\t{}\t{}\n".format(fragment, chrA, chrB, "\t".join(map(str, fields))))
                if splits[chrA][chrB]:
                    split_rows.append((chrA, chrB, splits[chrA][chrB]))
    WRITTEN_TABLES.clear()
    WRITTEN_TABLES[(os.path.abspath(disc_path), os.path.abspath(split_path))] = (_file_stamp(disc_path), _file_stamp(split_path), disc_rows, split_rows)


def _file_stamp(path):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns, st.st_ino)


def written_tables(disc_path, split_path):
    """the (discordant, split) rows of the last `main` of THIS process if the two files on disk are still the ones it wrote
    (size, mtime, inode) — else None, and the caller parses the text.  Per contig pair in file order: discordants
    (chrA, chrB, [(fragment, chrA, chrB, fields as written, not yet str), ...]), splits (chrA, chrB, {fragment: fields})."""
    ent = WRITTEN_TABLES.get((os.path.abspath(disc_path), os.path.abspath(split_path)))
    if ent is None:
        return None
    try:
        if _file_stamp(disc_path) != ent[0] or _file_stamp(split_path) != ent[1]:
            return None
    except OSError:
        return None
    return ent[2], ent[3]


def _main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len):
    t = time.time()
    header, chromosomes, coverage_data, res_data, res_splits, res_clips = scan_signals(
        bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, 50)
    STAGE_SECONDS.clear()
    STAGE_SECONDS["scan (ingest, coverage, predicates, rows)"] = time.time() - t
    STAGE_SECONDS.update({"  " + k: v for k, v in SCAN_SECONDS.items()})
    for hook in list(AFTER_SCAN):
        hook()
    t1 = time.time()
    _merge_and_write(header, chromosomes, res_data, res_splits, res_clips, prefix, sample_id)
    print("total", time.time() - t)
    STAGE_SECONDS["merge + write .tab / clips"] = time.time() - t1
    t1 = time.time()
    del res_data, res_splits, res_clips
    STAGE_SECONDS["free the row lists"] = time.time() - t1
    STAGE_SECONDS["_end"] = time.time()
    return coverage_data


def _main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group):
    """tiddit_signal.main with one process per GPU on ONE file (BASELINE configs[4]).  The reference fans out one worker per contig
    and merges their rows in contig order (:259-284); here rank r scans the records that start in its 1/N of the file's bytes
    (BGZF blocks are independent; the seams are checked, dist.check_seams), the 50-bp bins meet in ONE exact all-reduce, and the
    rows of every contig are gathered on rank 0 in RANK order — the file is coordinate sorted, so that is the file order of the
    single-process scan, and the per-fragment merge (a fragment's two reads may sit on different ranks) runs after the ordered
    gather exactly as in :262-284.  Rank 0 writes the byte-identical .tab / clip files; every rank returns the coverage dictionary."""
    import pickle
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

    header, chromosomes, coverage_data, res_data, res_splits, res_clips = scan_signals(
        bam_file_name, min_q, max_ins, min_contig, min_anchor_len, min_clip_len, 50, shard=(rank, world), reduce_bins=reduce_bins)
    STAGE_SECONDS.clear()
    STAGE_SECONDS["scan (ingest, coverage, predicates, rows; this rank's shard)"] = time.time() - t
    STAGE_SECONDS.update({"  " + k: v for k, v in SCAN_SECONDS.items()})
    for hook in list(AFTER_SCAN):
        hook()
    t1 = time.time()
    mine = {c: (res_data[c], res_splits[c], [_clip_bytes(x) for x in res_clips[c]]) for c in chromosomes
            if res_data[c] or res_splits[c] or res_clips[c]}
    parts = tdist.gather_bytes(pickle.dumps(mine, protocol=4), 0, group)
    STAGE_SECONDS["row gather"] = time.time() - t1
    t1 = time.time()
    if rank == 0:
        parts = [pickle.loads(p) for p in parts]
        data = {c: [] for c in chromosomes}
        splits = {c: [] for c in chromosomes}
        clips = {c: [] for c in chromosomes}
        for part in parts:                                   # rank order = file order inside every contig
            for c, (d, s, cl) in part.items():
                data[c] += d
                splits[c] += s
                clips[c] += [[x, ""] for x in cl]
        _merge_and_write(header, chromosomes, data, splits, clips, prefix, sample_id)
    dist.barrier(group)                                      # the files exist when any rank returns
    STAGE_SECONDS["merge + write .tab / clips"] = time.time() - t1
    return coverage_data


def main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group=None):
    """:func:`main` with one process per GPU (torch.distributed initialised; nccl = RCCL, or gloo) — see :func:`_main_sharded`"""
    with quiet_gc():
        return _main_sharded(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len, group)


def main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len):
    """``tiddit_signal.main`` (tiddit_signal.pyx:230-334): signals of every contig -> discordants_/splits_ .tab, clips_ .fa; returns the
    50-bp coverage dictionary.  (The collector is off while the row tables are built: hostutil.quiet_gc.)"""
    with quiet_gc():
        coverage_data = _main(bam_file_name, ref, prefix, min_q, max_ins, sample_id, threads, min_contig, skip_index, min_anchor_len, min_clip_len)
    STAGE_SECONDS["collector back on"] = time.time() - STAGE_SECONDS.pop("_end")
    return coverage_data
