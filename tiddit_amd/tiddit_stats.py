"""Drop-in for ``tiddit.tiddit_stats.statistics`` (tiddit_stats.py:5-78): library statistics from the
first ``n_reads`` placed alignments — mean read length, insert-size mean / std / 99.9th percentile and
the pair-orientation vote.  Same sampling rules, evaluated on the decoded arrays instead of per read."""
import time

import numpy

from .bamio import open_bam


def statistics(bam_file_name, ref, min_mapq, max_ins_len, n_reads):
    library = {}
    t = time.time()
    reader = open_bam(bam_file_name)
    read_length, insert_size = [], []
    is_innie = is_outtie = 0
    n_sampled = 0
    for b in reader.batches():
        placed = numpy.flatnonzero(b.tid >= 0)            # samfile.fetch() skips the unplaced tail (:17)
        room = n_reads + 1 - n_sampled                    # the read that trips `n_sampled > n_reads` still adds its length (:19-23)
        take = placed[:room]
        read_length.append(b.l_seq[take])
        n_sampled += len(take)
        use = take[:max(0, min(len(take), n_reads - (n_sampled - len(take))))]
        flag = b.flag[use].astype(numpy.int32)
        tlen = b.tlen[use]
        ok = (flag & 0x8) == 0                                           # mate mapped (:25)
        ok &= ((flag & 0x10) != 0) != ((flag & 0x20) != 0)               # opposite strands (:28)
        ok &= (b.mate_tid[use] == b.tid[use]) & (tlen <= max_ins_len)    # same contig, not too far (:31)
        ok &= b.mate_pos[use] >= b.pos[use]                              # leftmost read of the pair (:34)
        ok &= ((flag & 0xd00) == 0) & (b.mapq[use] >= min_mapq)          # primary, not duplicate (:37)
        insert_size.append(tlen[ok])
        rev = (flag[ok] & 0x10) != 0
        is_outtie += int(rev.sum())                                      # reverse read first: outtie (:42-45)
        is_innie += int((~rev).sum())
        if n_sampled > n_reads:
            break
    reader.close()
    read_length = numpy.concatenate(read_length) if read_length else numpy.zeros(0)
    insert_size = numpy.concatenate(insert_size) if insert_size else numpy.zeros(0)
    library["avg_read_length"] = numpy.average(read_length)
    if len(insert_size):
        library["avg_insert_size"] = numpy.average(insert_size)
        library["std_insert_size"] = numpy.std(insert_size)
        library["percentile_insert_size"] = numpy.percentile(insert_size, 99.9)
    else:
        library["avg_insert_size"] = 0
        library["std_insert_size"] = 0
        library["percentile_insert_size"] = 0
    print("LIBRARY STATISTICS")
    if is_innie > is_outtie:
        library["mp"] = False
        print("\tPair orientation = Forward-Reverse")
    else:
        print("\tPair orientation = Reverse-Forward")
        library["mp"] = True
    print("\tAverage Read length = {}".format(library["avg_read_length"]))
    print("\tAverage insert size = {}".format(library["avg_insert_size"]))
    print("\tStdev insert size = {}".format(library["std_insert_size"]))
    print("\t99.95 percentile insert size = {}".format(library["percentile_insert_size"]))
    print("Calculated statistics in: " + str(t - time.time()))
    print("")
    return library
