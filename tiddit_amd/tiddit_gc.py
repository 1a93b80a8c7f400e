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


def binned_gc(fasta_path, contig, bin_size, n_cutoff):
    fasta = fasta_path if isinstance(fasta_path, FastaFile) else FastaFile(fasta_path)
    return [contig, binned_gc_array(fasta.fetch_array(contig), bin_size, n_cutoff)]


def main(reference, contigs, threads, bin_size, n_cutoff):
    fasta = FastaFile(reference)
    gc_dictionary = {}
    for contig in contigs:
        gc = binned_gc(fasta, contig, bin_size, n_cutoff)
        gc_dictionary[gc[0]] = gc[1]
    return gc_dictionary
