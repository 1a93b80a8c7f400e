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
