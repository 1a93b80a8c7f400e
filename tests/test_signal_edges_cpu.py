"""Edge vectors of the split-read analysis (tiddit_signal.pyx:11-145): the product's SA_analysis / find_SA_query_range against the
statement-by-statement restatement in oracle/signal_oracle.py.  PARITY UNPINNED (tiddit_signal.pyx cannot be compiled here: it
cimports pysam) — these vectors at least pin the two independent readings of the code to each other on the cases SURVEY §8 lists:
several SA entries (only entry 0 is ever used, :36-39), CIGAR letters outside MSHDI (KeyError, :23), string order of contig
names (`chr10 < chr2`, :110), reverse-strand reads and SA alignments, clips on either side."""
import itertools

import pytest

from oracle import signal_oracle
from tiddit_amd import tiddit_signal


class R:
    def __init__(self, qname, start, end, rev, qstart, sa):
        self.query_name, self.reference_start, self.reference_end, self.is_reverse, self.query_alignment_start = qname, start, end, rev, qstart
        self.tags = {"SA": sa}

    def get_tag(self, tag):
        return self.tags[tag]


def both(read, min_q, chrom):
    return tiddit_signal.SA_analysis(read, min_q, "SA", chrom), signal_oracle._SA_analysis(read, min_q, chrom)


@pytest.mark.parametrize("rev,strand,clip_first,sa_chr", list(itertools.product([False, True], "+-", [False, True], ["chr1", "chr10", "chr2", "chrX"])))
def test_orientation_grid(rev, strand, clip_first, sa_chr):
    for chrom in ("chr2", "chr10", "chr1"):
        for sa_pos in (500, 1500, 99000):
            cig = "60S90M" if clip_first else "90M60S"
            read = R("q1", 1000, 1090, rev, 0 if clip_first else 60, "%s,%d,%s,%s,60,0;" % (sa_chr, sa_pos, strand, cig))
            got, want = both(read, 5, chrom)
            assert list(got) == list(want) and len(got) == 11


def test_several_sa_entries_use_entry_zero():
    # entry 0 has mapq 3 < min_q -> the read is dropped although entry 1 would pass; and the other way round
    for sa, kept in (("chr2,500,+,70S80M,3,0;chr3,900,-,80M70S,60,1;", False), ("chr2,500,+,70S80M,60,0;chr3,900,-,80M70S,0,1;", True),
                     ("chr2,500,+,70S80M,60,0;chr3,900,-,80M70S,60,1;chr4,5,+,10M140S,60,9;", True)):
        got, want = both(R("q", 100, 250, False, 0, sa), 5, "chr1")
        assert list(got) == list(want) and bool(got) == kept
        if kept:
            assert got[1] == "chr2" and got[5] == 500


@pytest.mark.parametrize("cigar", ["50M10N90M", "150=", "70X80M", "20S100M30P"])
def test_cigar_letters_outside_MSHDI_raise_keyerror_like_the_reference(cigar):
    read = R("q", 100, 250, False, 0, "chr2,500,+,%s,60,0;" % cigar)
    with pytest.raises(KeyError):
        tiddit_signal.SA_analysis(read, 5, "SA", "chr1")
    with pytest.raises(KeyError):
        signal_oracle._SA_analysis(read, 5, "chr1")


def test_contig_names_compare_as_strings():
    # "chr10" < "chr2": the SA side becomes chrA although contig 2 comes first in the header
    got, want = both(R("q", 100, 250, False, 0, "chr10,7000,+,70S80M,60,0;"), 5, "chr2")
    assert list(got) == list(want) and got[0] == "chr10" and got[1] == "chr2"
    got, want = both(R("q", 100, 250, False, 0, "chr2,7000,+,70S80M,60,0;"), 5, "chr10")
    assert list(got) == list(want) and got[0] == "chr10" and got[1] == "chr2"


def test_hard_clips_deletions_insertions_in_sa_cigar():
    for cig in ("30H40S80M", "80M5D20M50S", "60S10I80M", "150M", "75S75M"):
        for qstart in (0, 40, 75):
            got, want = both(R("q", 2000, 2150, True, qstart, "chr1,1200,-,%s,30,2;" % cig), 5, "chr1")
            assert list(got) == list(want)
