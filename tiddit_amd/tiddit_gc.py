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


def main(reference, contigs, threads, bin_size, n_cutoff):
    fasta = FastaFile(reference)
    gc_dictionary = {}
    for contig in contigs:
        gc = binned_gc(fasta, contig, bin_size, n_cutoff)
        gc_dictionary[gc[0]] = gc[1]
    return gc_dictionary
