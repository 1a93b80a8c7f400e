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


def region_counts(table, chrom, starts, ends, bps, min_q, max_ins, ctx=None):
    """-> int64[nq, 7]: bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r for every (start, end, bp)"""
    ctx = ctx or _native.default_context()
    t = table.tid[chrom]
    a = table.contigs[t]
    qs = numpy.ascontiguousarray(starts, dtype=numpy.int32)
    qe = numpy.ascontiguousarray(ends, dtype=numpy.int32)
    qb = numpy.ascontiguousarray(bps, dtype=numpy.int32)
    out = numpy.zeros((len(qs), 7), dtype=numpy.int64)
    _native.check(ctx.lib.tdt_region_counts(ctx.handle, _native.ptr(a["start"]), _native.ptr(a["end"]), _native.ptr(a["mapq"]),
                                            _native.ptr(a["flag"]), _native.ptr(a["mate_tid"]), _native.ptr(a["mate_pos"]),
                                            _native.ptr(a["tlen"]), _native.ptr(a["has_sa"]), len(a["start"]), t, table.lengths[t],
                                            _native.ptr(qs), _native.ptr(qe), _native.ptr(qb), len(qs), int(min_q), int(max_ins),
                                            _native.ptr(out)))
    return out


def get_region(table, chrom, start, end, bp, min_q, max_ins, contig_number=None):
    """One candidate, the reference's return value (tiddit_variant.pyx:141-151)."""
    bases, n_reads, low_q, n_discs, n_splits, crossing_f, crossing_r = (int(v) for v in region_counts(table, chrom, [start], [end], [bp],
                                                                                                     min_q, max_ins)[0])
    coverage = bases / (end - start + 1)
    frac_low_q = low_q / float(n_reads) if n_reads > 0 else 0
    return (coverage, frac_low_q, n_discs, n_splits, crossing_f, crossing_r)
