"""Host-side format code (CPU): BGZF/BAM writer -> C record decoder round trip, checked against the
independent pure-Python parser of oracle/signal_oracle.py; FASTA index reader."""
import numpy as np
import pytest

from oracle import signal_oracle
from tiddit_amd import bamio, build, synth, synth_bam
from tiddit_amd.fasta import FastaFile


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    build.build()
    p = str(tmp_path_factory.mktemp("bam") / "syn.bam")
    info = synth_bam.write_synthetic_bam(p, [("chr1", 60000), ("chr2", 40000), ("chrM", 3000), ("tiny", 500)], depth=6, seed=3)
    return p, info


def test_bam_roundtrip_against_independent_parser(bam):
    path, info = bam
    hdr, reads = signal_oracle.parse_bam(path)
    rd = bamio.BamReader(path, batch_bytes=200_000)      # small batches: records straddle batch boundaries
    assert rd.header["SQ"] == hdr["SQ"] and rd.header["RG"][0]["SM"] == "SYN"
    k = 0
    nb = 0
    for b in rd.batches():
        nb += 1
        for i in range(len(b)):
            r = reads[k]
            assert (b.tid[i], b.pos[i], b.end[i], b.mapq[i], b.flag[i], b.mate_tid[i], b.tlen[i]) == \
                (r.reference_id, r.reference_start, r.reference_end, r.mapq, r.flag, r.next_reference_id, r.isize), k
            assert (int(b.sa_off[i]) >= 0) == ("SA" in r.tags)
            if k % 37 == 0 or "SA" in r.tags:
                v = b.record(i)
                assert v.query_name == r.query_name and v.cigartuples == r.cigartuples and v.query_sequence == r.query_sequence
                if "SA" in r.tags:
                    assert v.get_tag_sa() == r.tags["SA"]
            if r.cigartuples:
                assert (int(b.cigar_first[i]) & 0xf, int(b.cigar_first[i]) >> 4) == r.cigartuples[0]
                assert (int(b.cigar_last[i]) & 0xf, int(b.cigar_last[i]) >> 4) == r.cigartuples[-1]
            else:
                assert b.cigar_first[i] == 0xffffffff
            k += 1
    assert k == len(reads) == info["n_records"] and nb > 3
    # coordinate sorted
    keys = [(r.reference_id, r.reference_start) for r in reads]
    assert keys == sorted(keys)


def test_fasta_reader(tmp_path):
    fa = tmp_path / "r.fa"
    seqs = {"a": synth.gen_sequence(1234, seed=1), "b": synth.gen_sequence(60, seed=2), "c": synth.gen_sequence(61, seed=3)}
    with open(fa, "w") as f:
        for n, s in seqs.items():
            f.write(">%s desc\n" % n)
            t = s.tobytes().decode()
            f.write("\n".join(t[i:i + 60] for i in range(0, len(t), 60)) + "\n")
    ff = FastaFile(str(fa))
    assert ff.references == ["a", "b", "c"]
    for n, s in seqs.items():
        assert ff.get_reference_length(n) == len(s)
        assert np.array_equal(ff.fetch_array(n), s)
    assert ff.fetch("a", 100, 150) == seqs["a"][100:150].tobytes().decode()


def test_native_bgzf_inflate_matches_zlib_and_checks_crc(bam, tmp_path):
    """tdt_bgzf_scan / tdt_bgzf_inflate (threaded host inflate) vs Python zlib block by block; corrupt and truncated input fail loudly"""
    import ctypes
    from tiddit_amd import _native
    path, _ = bam
    lib = _native.load()
    comp = np.fromfile(path, dtype=np.uint8)
    want = b"".join(bamio.bgzf_blocks(open(path, "rb")))
    nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
    assert consumed.value == len(comp) and produced.value == len(want) and nb.value > 3
    for threads in (1, 3, 0):
        out = np.zeros(produced.value, dtype=np.uint8)
        _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(out), len(out), threads))
        assert out.tobytes() == want
    # the scan stops at max_out and at a partial trailing block
    _native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp) - 5, 100_000, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
    assert 0 < produced.value <= 100_000 and nb.value >= 1
    part = np.zeros(produced.value, dtype=np.uint8)
    _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(part), len(part), 2))
    assert part.tobytes() == want[:produced.value]
    # one flipped payload byte -> CRC/inflate failure, never silent
    bad = comp.copy()
    bad[consumed.value // 2] ^= 0x55
    with pytest.raises(_native.TdtError):
        _native.check(lib.tdt_bgzf_inflate(_native.ptr(bad), consumed.value, _native.ptr(part), len(part), 2))
    with pytest.raises(_native.TdtError):                                    # wrong output size
        _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(part), len(part) - 1, 2))
    with pytest.raises(_native.TdtError):                                    # not BGZF
        _native.check(lib.tdt_bgzf_scan(_native.ptr(part), len(part), 1 << 30, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
    trunc = str(tmp_path / "trunc.bam")
    comp[:len(comp) - 40].tofile(trunc)                                       # cut inside the last data block (EOF marker is 28 bytes)
    with pytest.raises(ValueError):
        for _ in bamio.BamReader(trunc).batches():
            pass
    assert lib.tdt_host_threads(0) >= 1


@pytest.mark.parametrize("seed", range(6))
def test_host_decoder_on_random_legal_records(tmp_path, seed):
    """records of every legal shape (all CIGAR operations, 1..250-character names, empty sequences, every aux type including arrays):
    the C decoder agrees with the independent pure-Python parser field by field"""
    build.build()
    path = str(tmp_path / "rnd.bam")
    n = synth_bam.write_random_bam(path, 500 + seed)
    hdr, reads = signal_oracle.parse_bam(path)
    rd = bamio.BamReader(path, batch_bytes=150_000)
    k = 0
    for b in rd.batches():
        for i in range(len(b)):
            r = reads[k]
            assert (b.tid[i], b.pos[i], b.end[i], b.mapq[i], b.flag[i], b.mate_tid[i], b.mate_pos[i], b.tlen[i], b.l_seq[i]) == \
                (r.reference_id, r.reference_start, r.reference_end, r.mapq, r.flag, r.next_reference_id, r.mate_pos, r.isize, len(r.query_sequence)), k
            assert (int(b.sa_off[i]) >= 0) == ("SA" in r.tags)
            if "SA" in r.tags:
                assert b.record(i).get_tag_sa() == r.tags["SA"]
            if k % 53 == 0:
                v = b.record(i)
                assert v.query_name == r.query_name and v.cigartuples == r.cigartuples
            k += 1
    rd.close()
    assert k == n == len(reads)
