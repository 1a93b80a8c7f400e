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


def test_native_split_fields_equal_the_literal_code():
    """tdt_split_fields (the numeric half of SA_analysis in C, what the device pipeline's row builder calls) + split_rows_native ==
    SA_analysis on real BAM records: the orientation grid, multi-entry tags, hard clips / deletions / insertions, hard-clipped reads,
    and unusual tags (other CIGAR letters, signed numbers, an odd strand) that must fall through to the literal code and raise"""
    import numpy as np
    from tiddit_amd import _native, bamio, build
    build.build()
    lib = _native.load()
    names = ["chr1", "chr10", "chr2", "chrX"]
    recs, expect_err = [], []
    k = 0
    for rev, strand, own_cig, sa_chr, sa_pos in itertools.product([False, True], "+-", ["60S90M", "90M60S", "5H20S125M", "150M"],
                                                                  ["chr1", "chr10", "chr2", "chrX", "chrUn"], [500, 1500, 99000]):
        for sa_cig, mq in (("70S80M", 60), ("80M70S", 4), ("30H40S80M", 60), ("80M5D20M50S", 60), ("60S10I80M", 60), ("75S75M", 60)):
            tid = k % 3
            recs.append(("q%d" % k, 0x10 if rev else 0, tid, 1000 + k, own_cig, "%s,%d,%s,%s,%d,0;chr2,7,+,150M,60,1;" % (sa_chr, sa_pos, strand, sa_cig, mq)))
            k += 1
    for bad in ("chr2,500,+,50M10N90M,60,0;", "chr2,500,+,150=,60,0;", "chr2,+500,+,150M,60,0;", "chr2,500,x,150M,60,0;", "chr2,500,+,150M,6_0,0;"):
        recs.append(("bad%d" % k, 0, 0, 5000 + k, "150M", bad))
        k += 1
    raw, meta, raw_end = bytearray(), [], []
    for q, flag, tid, pos, cig, sa in recs:
        r = bamio.encode_record(q, flag, tid, pos, 60, cig, tid, pos + 300, 450, seq="A" * 150, tags=[("SA", "Z", sa)])
        ref = sum(l for op, l in bamio.parse_cigar(cig) if op in (0, 2))
        meta.append((len(meta), tid, pos, pos + ref, tid, r.index(b"SAZ") + 3, flag, 4, 0))
        raw += r
        raw_end.append(len(raw))

    class Sel:
        pass
    sel = Sel()
    sel.meta = np.array(meta, dtype=tiddit_signal._META)
    sel.raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    sel.raw_bytes = bytes(raw)
    sel.raw_end = np.array(raw_end, dtype=np.uint32)
    sel.rec_off = np.concatenate([[0], sel.raw_end[:-1]]).astype(np.uint64)
    sel.sa_off = sel.rec_off.astype(np.int64) + sel.meta["sa_rel"]
    sel.tid, sel.pos, sel.end, sel.flag = sel.meta["tid"], sel.meta["pos"], sel.meta["end"], sel.meta["flag"]
    sel.record = lambda i: bamio.RecordView(sel, i)
    good = np.arange(len(recs) - 5)
    got = {n: [] for n in names}
    tiddit_signal.split_rows_native(sel, good, names, 5, got, lib=lib)
    want = {n: [] for n in names}
    for i in good:
        row = tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, int(i)), 5, "SA", names[sel.tid[i]])
        if row:
            want[names[sel.tid[i]]].append(list(row))
    assert got == want and sum(len(v) for v in got.values()) > 500
    assert any(r[0] != n for n in names for r in got[n]) and any(r[0] == n and r[1] != n for n in names for r in got[n])
    so = np.empty(len(recs), dtype=tiddit_signal._SPLIT)
    _native.check(lib.tdt_split_fields(_native.ptr(sel.meta), _native.ptr(sel.raw_end), _native.ptr(sel.raw), len(sel.raw),
                                       _native.ptr(np.arange(len(recs), dtype=np.uint32)), len(recs), 5, _native.ptr(so)))
    assert set(so["status"][:len(good)].tolist()) == {0, 1} and (so["status"][len(good):] == 2).all()
    for i in range(len(good), len(recs)):                  # unusual tags: the literal code decides (KeyError / ValueError / a row)
        a = b = None
        try:
            a = tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, i), 5, "SA", "chr1")
        except Exception as e:
            a = type(e)
        try:
            g = {n: [] for n in names}
            tiddit_signal.split_rows_native(sel, np.array([i]), names, 5, g, lib=lib)
            b = g["chr1"][0] if g["chr1"] else ()
        except Exception as e:
            b = type(e)
        assert (list(a) if isinstance(a, (list, tuple)) else a) == (list(b) if isinstance(b, (list, tuple)) else b), recs[i]
