"""Drop-in for ``tiddit.tiddit_coverage_analysis.determine_ploidy`` (tiddit_coverage_analysis.pyx:9-41):
per-contig median of the coverage bins that are covered and not N-masked, genome-wide median, ploidy table.
The reference walks every bin in Python and calls numpy.median on the survivors; here the masked medians of all
contigs and of the whole genome come from ONE device call (radix select on the float64 bit patterns,
csrc/tdt_median.hip); the two middle values are averaged with numpy exactly like numpy.median does."""
import numpy

from . import _native


def masked_medians(pairs, ctx=None):
    """pairs: list of (coverage float64[], gc int8[]) -> (list of per-pair medians, median over all pairs).
    median of { cov[i] : cov[i] > 0 and gc[i] != -1 }; nan for an empty selection (numpy.median([]))."""
    import ctypes
    ctx = ctx or _native.default_context()
    covs, gcs = [], []
    for cov, gc in pairs:
        n = len(cov)
        if len(gc) < n:
            raise IndexError("gc array shorter than its coverage array")     # the reference indexes gc[chromosome][i]
        covs.append(numpy.ascontiguousarray(cov, dtype=numpy.float64))
        gcs.append(numpy.ascontiguousarray(gc[:n], dtype=numpy.int8))
    # every contig's arrays go to the device from where they lie (tdt_masked_medians_parts): joining them on the host first was a
    # 0.5-GB copy for a human genome's 60 M bins
    k = len(covs)
    nseg = k + 1
    cov_ptrs = (ctypes.c_void_p * max(k, 1))(*[c_.ctypes.data for c_ in covs])
    gc_ptrs = (ctypes.c_void_p * max(k, 1))(*[g_.ctypes.data for g_ in gcs])
    lens = numpy.array([len(c_) for c_ in covs], dtype=numpy.int64)
    lower, upper = numpy.empty(nseg), numpy.empty(nseg)
    count = numpy.empty(nseg, dtype=numpy.int64)
    _native.check(ctx.lib.tdt_masked_medians_parts(ctx.handle, cov_ptrs, gc_ptrs, _native.ptr(lens) if k else None, k,
                                                   _native.ptr(lower), _native.ptr(upper), _native.ptr(count)))
    med = [numpy.mean([lower[s], upper[s]]) if count[s] else numpy.nan for s in range(nseg)]
    return med[:-1], med[-1]


def determine_ploidy(coverage_data, contigs, library, ploidy, prefix, c, reference_fasta, bin_size, bam_header, gc):
    f = open("{}.ploidies.tab".format(prefix), "w")
    f.write("Chromosome\tPloidy\tPloidy_rounded\tMean_coverage\n")
    names = list(coverage_data)
    per_contig, overall = masked_medians([(coverage_data[ch], gc[ch]) for ch in names])
    for chromosome, med in zip(names, per_contig):
        library["avg_coverage_{}".format(chromosome)] = 0 if numpy.isnan(med) else med
    library["avg_coverage"] = c if c else overall
    for chromosome in contigs:
        if chromosome not in coverage_data:
            continue
        avg = library["avg_coverage_{}".format(chromosome)]
        library["contig_ploidy_{}".format(chromosome)] = int(round(ploidy * avg / library["avg_coverage"]))
        f.write("{}\t{}\t{}\t{}\n".format(chromosome, avg / library["avg_coverage"] * ploidy,
                                          library["contig_ploidy_{}".format(chromosome)], avg))
    f.close()
    return library
