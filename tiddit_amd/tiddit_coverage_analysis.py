"""Drop-in for ``tiddit.tiddit_coverage_analysis.determine_ploidy`` (tiddit_coverage_analysis.pyx:9-41):
per-contig median of the coverage bins that are covered and not N-masked, genome median, ploidy table.
The reference walks every bin in Python; here the mask is one numpy expression per contig (host-side
consumer of the two device histograms — SURVEY §8(f) row 1)."""
import numpy


def determine_ploidy(coverage_data, contigs, library, ploidy, prefix, c, reference_fasta, bin_size, bam_header, gc):
    f = open("{}.ploidies.tab".format(prefix), "w")
    f.write("Chromosome\tPloidy\tPloidy_rounded\tMean_coverage\n")
    all_cov = []
    for chromosome in coverage_data:
        cov = coverage_data[chromosome]
        g = gc[chromosome]
        n = len(cov)
        keep = cov[(cov > 0) & (numpy.asarray(g[:n]) != -1)] if len(g) >= n else None
        if keep is None:
            raise IndexError("gc array shorter than the coverage array of " + chromosome)
        all_cov.append(keep)
        med = numpy.median(keep) if len(keep) else numpy.nan
        library["avg_coverage_{}".format(chromosome)] = 0 if numpy.isnan(med) else med
    if not c:
        flat = numpy.concatenate(all_cov) if all_cov else numpy.zeros(0)
        library["avg_coverage"] = numpy.median(flat) if len(flat) else numpy.nan
    else:
        library["avg_coverage"] = c
    for chromosome in contigs:
        if chromosome not in coverage_data:
            continue
        avg = library["avg_coverage_{}".format(chromosome)]
        library["contig_ploidy_{}".format(chromosome)] = int(round(ploidy * avg / library["avg_coverage"]))
        f.write("{}\t{}\t{}\t{}\n".format(chromosome, avg / library["avg_coverage"] * ploidy,
                                          library["contig_ploidy_{}".format(chromosome)], avg))
    f.close()
    return library
