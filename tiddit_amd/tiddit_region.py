"""Regional evidence counts for SV candidates on the MI355X — the hot loop of ``tiddit_variant.get_region``
(tiddit_variant.pyx:54-151).  The reference re-opens the BAM region of every candidate through pysam; here a
contig's decoded records are kept as packed arrays (:class:`ReadTable`) and all candidates of the contig are
answered by one kernel launch, one wavefront per candidate (csrc/tdt_region.hip).

``get_region(table, chr, start, end, bp, min_q, max_ins)`` returns the reference's 6-tuple
``(coverage, frac_low_q, n_discs, n_splits, crossing_f, crossing_r)``; ``region_counts`` is the batch form.
"""
import numpy

from . import _native
from .bamio import open_bam

_FIELDS = (("start", "pos", numpy.int32), ("end", "end", numpy.int32), ("mapq", "mapq", numpy.uint8), ("flag", "flag", numpy.uint16),
           ("mate_tid", "mate_tid", numpy.int32), ("mate_pos", "mate_pos", numpy.int32), ("tlen", "tlen", numpy.int32))


class ReadTable:
    """Per-contig packed alignment records of a coordinate-sorted BAM (what a region fetch iterates over)."""

    def __init__(self, bam_file_name):
        reader = open_bam(bam_file_name)
        self.references, self.lengths = reader.references, reader.lengths
        self.tid = {n: i for i, n in enumerate(self.references)}
        parts = {i: {k: [] for k, _, _ in _FIELDS} for i in range(len(self.references))}
        for i in parts:
            parts[i]["has_sa"] = []
        for b in reader.batches():
            edges = numpy.flatnonzero(numpy.diff(b.tid)) + 1
            for lo, hi in zip(numpy.concatenate([[0], edges]), numpy.concatenate([edges, [len(b)]])):
                t = int(b.tid[lo])
                if t < 0:
                    continue
                for k, src, _ in _FIELDS:
                    parts[t][k].append(getattr(b, src)[lo:hi])
                parts[t]["has_sa"].append((b.sa_off[lo:hi] >= 0).astype(numpy.uint8))
        reader.close()
        self.contigs = {}
        for t, d in parts.items():
            self.contigs[t] = {k: (numpy.concatenate(v) if v else numpy.zeros(0, dtype=dt))
                               for (k, _, dt), v in zip(_FIELDS, (d[k] for k, _, _ in _FIELDS))}
            self.contigs[t]["has_sa"] = numpy.concatenate(d["has_sa"]) if d["has_sa"] else numpy.zeros(0, dtype=numpy.uint8)


def _resident(table, t, ctx):
    """the contig's arrays in HBM (uploaded on first use, kept for the life of the table) -> (dict of tensors, max read span)"""
    import torch
    cache = table.__dict__.setdefault("_device", {})
    if t not in cache:
        a = table.contigs[t]
        dev = torch.device("cuda", ctx.device)
        ten = {k: torch.from_numpy(v.view(numpy.int16) if v.dtype == numpy.uint16 else v).to(dev) for k, v in a.items()}
        if len(a["start"]) and numpy.any(numpy.diff(a["start"]) < 0):
            raise ValueError("tiddit_region: reads of %s are not coordinate sorted" % table.references[t])
        span = int((a["end"].astype(numpy.int64) - a["start"]).max()) if len(a["start"]) else 1
        torch.cuda.synchronize(dev)
        cache[t] = (ten, max(1, span))
    return cache[t]


def region_counts(table, chrom, starts, ends, bps, min_q, max_ins, ctx=None):
    """-> int64[nq, 7]: bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r for every (start, end, bp).
    The contig's reads stay resident on the device between calls; only the candidates travel."""
    import torch
    ctx = ctx or _native.default_context()
    t = table.tid[chrom]
    ten, span = _resident(table, t, ctx)
    dev = ten["start"].device
    nq = len(starts)
    q = torch.from_numpy(numpy.ascontiguousarray(numpy.stack([numpy.asarray(starts), numpy.asarray(ends), numpy.asarray(bps)]), dtype=numpy.int32)).to(dev)
    out = torch.empty((nq, 7), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)                       # torch's copies run on its stream, the library on its own
    n = int(ten["start"].numel())
    _native.check(ctx.lib.tdt_region_counts_device(ctx.handle, ten["start"].data_ptr(), ten["end"].data_ptr(), ten["mapq"].data_ptr(),
                                                   ten["flag"].data_ptr(), ten["mate_tid"].data_ptr(), ten["mate_pos"].data_ptr(),
                                                   ten["tlen"].data_ptr(), ten["has_sa"].data_ptr(), n, t, span, table.lengths[t],
                                                   q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), nq, int(min_q), int(max_ins),
                                                   out.data_ptr()))
    ctx.sync()
    return out.cpu().numpy()


def get_region(table, chrom, start, end, bp, min_q, max_ins, contig_number=None):
    """One candidate, the reference's return value (tiddit_variant.pyx:141-151)."""
    bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r = (int(v) for v in region_counts(table, chrom, [start], [end], [bp],
                                                                                                     min_q, max_ins)[0])
    coverage = bases / (end - start + 1)
    frac_low_q = low_q / float(n_reads) if n_reads > 0 else 0
    return (coverage, frac_low_q, n_discs, n_splits, crossing_f, crossing_r)


# ------------------------------------------------------------------------------------------------------------------
# Region means of the coverage bins (tiddit_variant.pyx:265-283, 307-315)
class BinTable:
    """The 50-bp coverage bins (and GC bins) of every contig, concatenated once — the arrays the segment-mean kernel indexes."""

    def __init__(self, coverage_data, gc=None):
        self.offset, self.length = {}, {}
        o = 0
        for name, bins in coverage_data.items():
            self.offset[name], self.length[name] = o, len(bins)
            o += len(bins)
        self.cov = numpy.ascontiguousarray(numpy.concatenate([numpy.asarray(b, dtype=numpy.float64) for b in coverage_data.values()])
                                           if coverage_data else numpy.zeros(0))
        self.gc = None
        if gc is not None:
            self.gc = numpy.ascontiguousarray(numpy.concatenate([numpy.asarray(gc[n][:len(coverage_data[n])], dtype=numpy.int8) for n in coverage_data]))
            if len(self.gc) != len(self.cov):
                raise IndexError("gc array shorter than its coverage array")

    def segment(self, chrom, s, e):
        """the index range Python's ``coverage_data[chrom][s:e]`` selects (non-negative s, e), as offsets into the concatenation"""
        n, o = self.length[chrom], self.offset[chrom]
        s, e = min(max(int(s), 0), n), min(max(int(e), 0), n)
        return o + s, o + max(s, e)


def region_means(table, segments, masked=None, ctx=None):
    """segments: list of (chrom, s, e) bin slices; masked[i] truthy: only bins with gc > -1 count (needs table.gc).
    -> (float64 means — numpy.average of the slice, bit for bit; nan for an empty one —, int64 counts of the bins averaged)"""
    ctx = ctx or _native.default_context()
    nq = len(segments)
    lo = numpy.empty(nq, dtype=numpy.int64)
    hi = numpy.empty(nq, dtype=numpy.int64)
    for i, (chrom, s, e) in enumerate(segments):
        lo[i], hi[i] = table.segment(chrom, s, e)
    m = numpy.zeros(nq, dtype=numpy.uint8) if masked is None else numpy.ascontiguousarray(masked, dtype=numpy.uint8)
    mean = numpy.empty(nq, dtype=numpy.float64)
    count = numpy.empty(nq, dtype=numpy.int64)
    _native.check(ctx.lib.tdt_segment_means(ctx.handle, _native.ptr(table.cov), _native.ptr(table.gc) if table.gc is not None else None,
                                            len(table.cov), _native.ptr(lo), _native.ptr(hi), _native.ptr(m), nq, _native.ptr(mean), _native.ptr(count)))
    return mean, count


def candidate_means(sv_clusters, coverage_data, gc, library, bin_size=50):
    """The three coverage means tiddit_variant.define_variant forms for every candidate (:265-283, :307-315), all candidates of all
    chromosome pairs in ONE device call: -> {(chrA, chrB, cluster): {"avg_a", "avg_b", "covM" (None where the reference takes it
    from get_region instead: breakpoints closer than 1000 bp; 0 for inter-chromosomal candidates)}}"""
    import math
    table = BinTable(coverage_data, gc)
    keys, segs, masked = [], [], []
    for chrA in sv_clusters:
        for chrB in sv_clusters[chrA]:
            for cid, c in sv_clusters[chrA][chrB].items():
                posA, posB = c["posA"], c["posB"]
                if chrA == chrB and posA > posB:
                    posA, posB = posB, posA
                keys.append((chrA, chrB, cid, posA, posB))
                segs.append((chrA, int(math.floor(c["startA"] / float(bin_size))), int(math.floor(c["endA"] / float(bin_size))) + 1))
                segs.append((chrB, int(math.floor(c["startB"] / float(bin_size))), int(math.floor(c["endB"] / float(bin_size))) + 1))
                between = chrA == chrB and abs(posB - posA) >= 1000
                segs.append((chrA, int(math.floor(posA / float(bin_size))), int(math.floor(posB / float(bin_size))) + 1) if between else (chrA, 0, 0))
                masked += [0, 0, 1 if between else 0]
    mean, count = region_means(table, segs, masked) if segs else (numpy.zeros(0), numpy.zeros(0, dtype=numpy.int64))
    out = {}
    for i, (chrA, chrB, cid, posA, posB) in enumerate(keys):
        if chrA != chrB:
            covM = 0
        elif abs(posB - posA) < 1000:
            covM = None
        else:
            covM = mean[3 * i + 2] if count[3 * i + 2] > 4 else library["avg_coverage_{}".format(chrA)]
        out[(chrA, chrB, cid)] = {"avg_a": mean[3 * i], "avg_b": mean[3 * i + 1], "covM": covM}
    return out
