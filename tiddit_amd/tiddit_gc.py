"""Drop-in for ``tiddit.tiddit_gc`` (tiddit_gc.pyx) on the MI355X.

``binned_gc(fasta_path, contig, bin_size, n_cutoff) -> [contig, int8[]]`` (:6-33) and
``main(reference, contigs, threads, bin_size, n_cutoff) -> dict`` (:35-42).  The per-character Python
loop is one HIP kernel (csrc/tdt_gc.hip); the joblib process fan-out is replaced by one process
streaming contigs through the device (``threads`` is accepted and ignored).
"""
import numpy

from . import _native
from .fasta import FastaFile


def binned_gc_array(seq, bin_size, n_cutoff, ctx=None):
    """uint8 sequence (ASCII bases) -> int8[ceil(len/bin_size)]"""
    ctx = ctx or _native.default_context()
    if isinstance(seq, (bytes, bytearray)):
        seq = numpy.frombuffer(seq, dtype=numpy.uint8)
    elif isinstance(seq, str):
        seq = numpy.frombuffer(seq.encode(), dtype=numpy.uint8)
    seq = numpy.ascontiguousarray(seq, dtype=numpy.uint8)
    bin_size = int(bin_size)
    if bin_size <= 0:
        raise ZeroDivisionError("bin_size must be positive")
    nbins = -(-len(seq) // bin_size)
    out = numpy.zeros(nbins, dtype=numpy.int8)
    _native.check(ctx.lib.tdt_gc_bins(ctx.handle, _native.ptr(seq), len(seq), bin_size, float(n_cutoff), _native.ptr(out)))
    return out


def binned_gc(fasta_path, contig, bin_size, n_cutoff, ctx=None):
    """tiddit_gc.pyx:6-33.  The contig's bytes go to the device as they are in the file; the kernel steps over the line ends."""
    fasta = fasta_path if isinstance(fasta_path, FastaFile) else FastaFile(fasta_path)
    bin_size = int(bin_size)
    if bin_size <= 0:
        raise ZeroDivisionError("bin_size must be positive")
    length, _, linebases, linewidth = fasta.index[contig]
    if 0 < bin_size <= 2048 and 0 < length < (1 << 31) and 0 < linebases <= linewidth <= linebases + 2:
        ctx = ctx or _native.default_context()
        raw, length, linebases, linewidth = fasta.fetch_raw(contig)
        out = numpy.zeros(-(-length // bin_size), dtype=numpy.int8)
        _native.check(ctx.lib.tdt_gc_bins_fasta(ctx.handle, _native.ptr(raw), len(raw), length, linebases, linewidth, bin_size,
                                                float(n_cutoff), _native.ptr(out)))
        return [contig, out]
    return [contig, binned_gc_array(fasta.fetch_array(contig), bin_size, n_cutoff, ctx)]


_MANY_MAX_CONTIG = 8 << 20        # contigs with at most this many FASTA bytes travel together ...
_MANY_MAX_BATCH = 96 << 20        # ... up to this many bytes per call


def _gc_many(fasta, group, bin_size, n_cutoff, ctx, out_dict):
    """`group` = [(contig, length, offset, linebases, linewidth, nbytes)]: ONE device call for all of them (``tdt_gc_bins_fasta_many``)"""
    n = len(group)
    raw_off = numpy.zeros(n, dtype=numpy.int64)
    out_off = numpy.zeros(n, dtype=numpy.int64)
    o = b = 0
    for i, g in enumerate(group):
        raw_off[i], out_off[i] = o, b
        o += (g[5] + 15) & ~15                                     # every contig's bytes start at a multiple of 16 (the kernel's chunk)
        b += -(-g[1] // bin_size)
    raw = numpy.zeros(o + 16, dtype=numpy.uint8)
    with open(fasta.path, "rb") as f:
        for i, g in enumerate(group):
            if g[5]:
                f.seek(g[2])
                f.readinto(memoryview(raw)[raw_off[i]:raw_off[i] + g[5]])
    out = numpy.zeros(max(b, 1), dtype=numpy.int8)
    cols = [numpy.array([g[k] for g in group], dtype=dt) for k, dt in ((5, numpy.int64), (1, numpy.int64), (3, numpy.int32), (4, numpy.int32))]
    _native.check(ctx.lib.tdt_gc_bins_fasta_many(ctx.handle, _native.ptr(raw), len(raw), n, _native.ptr(raw_off), _native.ptr(cols[0]), _native.ptr(cols[1]),
                                                 _native.ptr(cols[2]), _native.ptr(cols[3]), int(bin_size), float(n_cutoff), _native.ptr(out),
                                                 _native.ptr(out_off), b))
    for i, g in enumerate(group):
        out_dict[g[0]] = out[out_off[i]:out_off[i] + -(-g[1] // bin_size)].copy()


def main(reference, contigs, threads, bin_size, n_cutoff):
    """tiddit_gc.pyx:35-42"""
    return gc_of_contigs(FastaFile(reference), contigs, bin_size, n_cutoff)


def gc_of_contigs(fasta, contigs, bin_size, n_cutoff, ctx=None):
    """{contig: int8 bins} in the order of `contigs`.  A human reference with its alt / decoy / HLA contigs has thousands of small
    contigs: they go to the device in groups (one call, one wait per group) instead of one round trip each."""
    bin_size = int(bin_size)
    if bin_size <= 0:
        raise ZeroDivisionError("bin_size must be positive")
    done, group, group_bytes = {}, [], 0
    ctx = ctx or _native.default_context()
    for contig in contigs:
        if contig in done:
            continue
        length, offset, linebases, linewidth = fasta.index[contig]
        direct = 0 < bin_size <= 2048 and 0 <= length < (1 << 31) and (length == 0 or 0 < linebases <= linewidth <= linebases + 2)
        nbytes = 0
        if direct and length:
            nfull = length // linebases
            tail = length - nfull * linebases
            nbytes = nfull * linewidth + tail - ((linewidth - linebases) if tail == 0 else 0)
        if not direct or nbytes > _MANY_MAX_CONTIG:
            done[contig] = binned_gc(fasta, contig, bin_size, n_cutoff, ctx=ctx)[1]
            continue
        done[contig] = None
        group.append((contig, length, offset, linebases, linewidth, nbytes))
        group_bytes += nbytes + 16
        if group_bytes >= _MANY_MAX_BATCH:
            _gc_many(fasta, group, bin_size, n_cutoff, ctx, done)
            group, group_bytes = [], 0
    if group:
        _gc_many(fasta, group, bin_size, n_cutoff, ctx, done)
    return {c: done[c] for c in contigs}
