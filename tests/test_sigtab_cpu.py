"""The native signal tables (csrc/tdt_sigtab.hip: host code behind the C ABI) without a GPU.  The device steps either side of them
are replaced by host stand-ins (tests/sigtab_common.py); everything between — clip entries, discordant / split rows, the merge of
tiddit_signal.main (tiddit_signal.pyx:246-284), the text of the two .tab files and the clip FASTA (:298-332), the signal table of
tiddit_cluster.main (tiddit_cluster.pyx:47-105) and the regrouping into candidates (:156-254) — is the product's code, compared with
  * the fixture the compiled reference / the restatements produced for the 3-Mb WGS-shaped file (tests/golden/sv_e2e_small.json),
  * the literal Python merge loop (tiddit_signal._merge_and_write) and the literal text parser + regrouping (tiddit_cluster._main),
  * oracle/signal_oracle.py and oracle/cluster_oracle.py on the same file.
PARITY of the BAM -> .tab stage itself stays UNPINNED (tiddit_signal.pyx cimports pysam): what is pinned here is that
the native tables and the literal Python give the same bytes, and from the .tab text downward the compiled reference's candidates."""
import hashlib
import os
import random

import numpy as np
import pytest

from oracle import cluster_oracle, signal_oracle
from tiddit_amd import bamio, tiddit_cluster, tiddit_signal
from tiddit_amd.sigtab import D_ROW, S_ROW, SignalTables

from sigtab_common import fill_tables, oracle_labels
from sv_e2e_common import load_fixture, materialise


def h(t):
    return hashlib.sha256(t if isinstance(t, bytes) else t.encode()).hexdigest()


def host_batches(bam, batch_bytes=None):
    rd = bamio.BamReader(bam, batch_bytes=batch_bytes)
    out = list(rd.batches())
    rd.close()
    return rd, out


@pytest.fixture(scope="module")
def small(golden_dir, tmp_path_factory):
    fx = load_fixture(golden_dir, "sv_e2e_small.json")
    d = str(tmp_path_factory.mktemp("sigtab_small"))
    bam, fa, contigs = materialise(fx, d, threads=4)
    rd, batches = host_batches(bam, batch_bytes=8 << 20)
    return fx, bam, contigs, rd, batches, d


def scan_args(fx):
    P = fx["params"]
    return P["min_q"], fx["library"]["percentile_insert_size"], P["min_anchor_len"], P["min_clip_len"]


def new_tables(rd, fx):
    return SignalTables(rd.references, rd.lengths, fx["params"]["min_contig"])


def literal_files(tables, header, prefix, sample):
    """the literal Python merge + writers over the rows the tables hold -> (discordants, splits, clips text)"""
    data, splits = tables.rows()
    names = tables.names
    clips = {n: ([[tables.clips(t), ""]] if tables.clips(t) else []) for t, n in enumerate(names)}
    chromosomes = [n for n, ln in zip(names, tables.lengths) if ln >= tables.min_contig]
    tiddit_signal._merge_and_write(header, chromosomes, data, splits, clips, prefix, sample)
    rd = lambda p: open(p, "rb").read()
    return rd("%s_tiddit/discordants_%s.tab" % (prefix, sample)), rd("%s_tiddit/splits_%s.tab" % (prefix, sample)), rd("%s_tiddit/clips_%s.fa" % (prefix, sample))


def test_tables_reproduce_the_fixture_and_the_literal_merge(small, tmp_path):
    fx, bam, contigs, rd, batches, d = small
    min_q, max_ins, anchor, clip = scan_args(fx)
    assert len(batches) > 3
    big = [ln >= fx["params"]["min_contig"] for ln in rd.lengths]
    t = new_tables(rd, fx)
    n_sel = fill_tables(t, batches, rd.references, big, min_q, max_ins, anchor, clip)
    st = t.stats()
    assert st["discordants_in_order"] and st["splits_in_order"] and st["discordant_rows"] > 1000 and st["split_rows"] > 200 and n_sel > 2000
    disc, dseg = t.text(0)
    split, sseg = t.text(1)
    assert h(disc) == fx["discordants_sha256"] and h(split) == fx["splits_sha256"]
    assert disc.count(b"\n") == fx["discordants_rows"] == int(dseg[:, 4].sum()) and split.count(b"\n") == fx["splits_rows"] == int(sseg[:, 4].sum())
    # segments: contig pairs in the order of main()'s nested dictionaries, back to back
    for seg, text in ((dseg, disc), (sseg, split)):
        assert [tuple(x) for x in seg[:, :2].tolist()] == sorted(tuple(x) for x in seg[:, :2].tolist())
        assert seg[0, 2] == 0 and (seg[1:, 2] == seg[:-1, 2] + seg[:-1, 3]).all() and seg[-1, 2] + seg[-1, 3] == len(text)
    kept = [i for i, ok in enumerate(big) if ok]
    clips_all = b"".join(t.clips(i) for i in kept)
    assert h(clips_all) == fx["clips_sha256"] and clips_all.count(b">") == fx["clips_entries"]
    # the literal Python merge over the very same rows writes the same three files
    ld, ls, lc = literal_files(t, rd.header, str(tmp_path / "lit"), "WGS")
    assert ld == disc and ls == split and lc == clips_all
    # ... and so does the restatement of tiddit_signal.main on the file
    cov, odisc, osplit, oclips, each, n = signal_oracle.signal_main_file(bam, min_q, max_ins, "WGS", fx["params"]["min_contig"], anchor, clip)
    assert odisc.encode() == disc and osplit.encode() == split and oclips.encode() == clips_all
    for i in kept:
        assert each[rd.references[i]].encode() == t.clips(i)
    # the block writer (what tiddit_signal.main runs): every file from per-contig blocks placed at their offsets
    prefix = str(tmp_path / "blk")
    tiddit_signal._write_tables(t, t, [rd.references[i] for i in kept], prefix, "WGS")
    assert open(prefix + "_tiddit/discordants_WGS.tab", "rb").read() == disc and open(prefix + "_tiddit/splits_WGS.tab", "rb").read() == split
    assert open(prefix + "_tiddit/clips_WGS.fa", "rb").read() == clips_all
    for i in kept:
        assert open(prefix + "_tiddit/clips/%s.fa" % rd.references[i], "rb").read() == t.clips(i)
    assert tiddit_signal.written_tables(prefix + "_tiddit/discordants_WGS.tab", prefix + "_tiddit/splits_WGS.tab") is t
    tiddit_signal._forget_tables()


@pytest.mark.parametrize("builder", ["cpython-api", "python-loop"])
@pytest.mark.parametrize("is_mp", [False, True])
def test_candidates_from_the_tables_equal_the_text_path_and_the_reference(small, tmp_path, monkeypatch, is_mp, builder):
    """tiddit_cluster.main twice on the same files — taking the native tables over, and parsing the text — with the device call
    replaced by the oracle's DBSCAN: the same nested dictionary, insertion order included; for the library's own orientation it is the
    compiled reference's (fixture), for the other one the restatement's"""
    fx, bam, contigs, rd, batches, d = small
    if builder == "python-loop":                            # the dictionaries by tiddit_cluster's own loop instead of tiddit_amd/_pycand
        monkeypatch.setenv("TIDDIT_PY_CANDIDATES", "1")
    elif tiddit_cluster._pycand is None:
        pytest.skip("tiddit_amd/_pycand was not built")
    P = fx["params"]
    min_q, max_ins, anchor, clip = scan_args(fx)
    big = [ln >= P["min_contig"] for ln in rd.lengths]
    t = new_tables(rd, fx)
    fill_tables(t, batches, rd.references, big, min_q, max_ins, anchor, clip)
    prefix = str(tmp_path / "c")
    tiddit_signal._write_tables(t, t, [n for n, ok in zip(rd.references, big) if ok], prefix, "WGS")
    monkeypatch.setattr(tiddit_cluster, "cluster_columns_device", oracle_labels)

    class Pageable:                                         # (pinned buffers need the HIP runtime's device: plain arrays here)
        def take(self, name, n, dtype):
            return np.zeros(n, dtype=dtype)
    monkeypatch.setattr(tiddit_cluster, "_POOL", Pageable())

    def buckets_by_oracle(buckets, epsilon, m, **kw):
        import oracle
        out = []
        for b in buckets:
            order = np.argsort(b[:, 0], kind="stable")
            lab = np.empty(len(b))
            lab[order] = oracle.dbscan_main(b[order], epsilon, m)
            out.append(lab)
        return out
    monkeypatch.setattr(tiddit_cluster, "cluster_buckets", buckets_by_oracle)
    names = [n for n, _ in contigs]
    args = (names, dict(contigs), ["WGS"], is_mp, fx["epsilon"], P["m"], max_ins, P["min_contig"], True, P["min_reads"])
    native = tiddit_cluster.main(prefix, *args)
    assert "signal table (native, into pinned columns)" in tiddit_cluster.STAGE_SECONDS
    tiddit_signal._forget_tables()
    text = tiddit_cluster.main(prefix, *args)
    assert "parse .tab" in tiddit_cluster.STAGE_SECONDS
    assert cluster_oracle.canonical(native) == cluster_oracle.canonical(text)
    assert [(a, list(native[a])) for a in native] == [(a, list(text[a])) for a in text]
    assert [list(native[a][b]) for a in native for b in native[a]] == [list(text[a][b]) for a in text for b in text[a]]     # insertion order of the ids
    assert cluster_oracle.canonical(native) == cluster_oracle.canonical(cluster_oracle.main(prefix, *args))
    if is_mp == fx["library"]["mp"]:
        assert cluster_oracle.summary(native) == fx["candidates"] and h(cluster_oracle.canonical(native)) == fx["candidates_sha256"]
    # same value TYPES in the two dictionaries (ints stay ints, orientation words stay strings)
    for a in native:
        for b in native[a]:
            for cid, c in native[a][b].items():
                o = text[a][b][cid]
                for side in ("positions_A", "positions_B"):
                    for k, v in c[side].items():
                        assert v == o[side][k] and [type(x) for x in v] == [type(x) for x in o[side][k]], (a, b, cid, side, k)
                assert type(c["posA"]) is type(o["posA"]) and type(c["startA"]) is type(o["startA"])


def test_batches_out_of_contig_order_fall_back_to_the_contig_major_merge(small, tmp_path):
    """an unsorted file: rows arrive with decreasing contig ids; the incremental merge stands down and finalize merges the log contig by
    contig — the result is the literal loop's over the per-contig lists in arrival order (what main() does with worker results)"""
    fx, bam, contigs, rd, batches, d = small
    min_q, max_ins, anchor, clip = scan_args(fx)
    big = [ln >= fx["params"]["min_contig"] for ln in rd.lengths]
    rng = random.Random(5)
    for trial in range(3):
        order = list(range(len(batches)))
        rng.shuffle(order)
        t = new_tables(rd, fx)
        fill_tables(t, [batches[i] for i in order], rd.references, big, min_q, max_ins, anchor, clip)
        st = t.stats()
        assert not st["discordants_in_order"]
        disc, _ = t.text(0)
        split, _ = t.text(1)
        ld, ls, lc = literal_files(t, rd.header, str(tmp_path / ("u%d" % trial)), "WGS")
        assert ld == disc and ls == split
        assert disc.count(b"\n") == fx["discordants_rows"]           # the same fragments pair up, whatever the order
        if trial == 0:
            assert h(disc) != fx["discordants_sha256"]               # ... in another order (first-seen order of the fragments)
        t.close()


def test_rows_shared_over_owner_ranks_give_the_single_table(small, tmp_path):
    """the N-rank job without processes: the batches are dealt to R scanning tables in file order, every table's rows are exported per
    owner rank of their chrA and imported in rank order by the owner — the owners' blocks, placed by size, are the single table's
    files, byte for byte; and a fragment whose two reads were scanned by different ranks still pairs up"""
    from tiddit_amd import dist as tdist
    fx, bam, contigs, rd, batches, d = small
    min_q, max_ins, anchor, clip = scan_args(fx)
    big = [ln >= fx["params"]["min_contig"] for ln in rd.lengths]
    one = new_tables(rd, fx)
    fill_tables(one, batches, rd.references, big, min_q, max_ins, anchor, clip)
    disc, _ = one.text(0)
    split, _ = one.text(1)
    for world in (2, 3, 5):
        cuts = [len(batches) * r // world for r in range(world + 1)]
        scanned = []
        for r in range(world):
            t = new_tables(rd, fx)
            fill_tables(t, batches[cuts[r]:cuts[r + 1]], rd.references, big, min_q, max_ins, anchor, clip)
            scanned.append(t)
        owner = tdist.contig_owners(rd.lengths, big, world)
        assert set(owner[[i for i, ok in enumerate(big) if ok]].tolist()) == set(range(world))
        merged = []
        for r in range(world):
            m = new_tables(rd, fx)
            for src in range(world):
                m.import_rows(scanned[src].export_rows(owner, r))
            merged.append(m)
        assert sum(m.stats()["discordant_rows"] for m in merged) <= one.stats()["discordant_rows"]      # (rows main() would skip are not sent)
        for what, want in ((0, disc), (1, split)):
            sizes = np.stack([m.sizes(what) for m in merged])
            assert ((sizes > 0).sum(axis=0) <= 1).all()                                                 # one owner per chrA
            path = str(tmp_path / ("w%d_%d" % (world, what)))
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
            total = sizes.sum(axis=0)
            base = np.cumsum(total) - total
            for r, m in enumerate(merged):
                for c in np.flatnonzero(sizes[r]):
                    m.pwrite(what, int(c), fd, int(base[c]))
            os.close(fd)
            assert open(path, "rb").read() == want, (world, what)
        for c in range(len(big)):
            assert b"".join(s.clips(c) for s in scanned) == one.clips(c)
        for t in scanned + merged:
            t.close()
    one.close()


CONTIGS = [("chr1", 50000), ("chr10", 40000), ("chr2", 30000), ("scaffoldA", 900), ("chrM", 16000)]


def _records(rows):
    """rows = [(qname, flag, tid, pos, cigar, mate_tid, tags, action)] -> (meta, raw_end, raw) as tdt_signal_scan_result returns them"""
    raw, meta, raw_end = bytearray(), [], []
    for q, flag, tid, pos, cig, mate, tags, action in rows:
        r = bamio.encode_record(q, flag, tid, pos, 60, cig, mate, pos + 300, 450, seq="ACGT" * 30, tags=tags)
        ref = sum(l for op, l in bamio.parse_cigar(cig) if op in (0, 2))
        sa_rel = r.index(b"SAZ") + 3 if b"SAZ" in r else -1
        meta.append((len(meta), tid, pos, pos + ref, mate, sa_rel, flag, action, 0))
        raw += r
        raw_end.append(len(raw))
    return np.array(meta, dtype=tiddit_signal._META), np.array(raw_end, dtype=np.uint32), np.frombuffer(bytes(raw), dtype=np.uint8)


def test_hand_built_reads_quirks_and_errors(tmp_path):
    names = [n for n, _ in CONTIGS]
    t = SignalTables(names, [ln for _, ln in CONTIGS], 1000)
    sa = lambda s: [("SA", "Z", s)]
    rows = [
        ("pairA", 0x1, 0, 100, "120M", 0, (), 8),                       # chr1 - chr1, forward
        ("pairB", 0x1 | 0x10, 0, 200, "120M", 2, (), 8),                # chr1 read, mate on chr2
        ("lonely", 0x1, 0, 300, "120M", 1, (), 8),                      # second read never arrives: not written
        ("clipA", 0x1, 0, 400, "40S80M", 0, (), 2),
        ("splitA", 0x1, 0, 500, "60S60M", 0, sa("chr10,7000,+,60M60S,60,0;"), 4),
        ("pairA", 0x1 | 0x10, 0, 900, "120M", 0, (), 8),
        ("triple", 0x1, 0, 1000, "120M", 0, (), 8),
        ("triple", 0x1 | 0x10, 0, 1100, "120M", 0, (), 8),
        ("triple", 0x1, 0, 1200, "120M", 0, (), 8),                     # a third read of the fragment: ignored by the writer (:300)
        ("splitA", 0x1 | 0x10, 1, 7000, "60M60S", 1, sa("chr1,501,+,60S60M,60,0;"), 4),      # the other read of the split fragment: fields appended (:282)
        ("tiny", 0x1, 1, 7100, "120M", 3, (), 8),                       # mate on scaffoldA (< min_contig): "chr10" < "scaffoldA" -> chrA = chr10, kept
        ("pairB", 0x1, 2, 250, "120M", 0, (), 8),                       # chr2 read of pairB: chrA = chr1 is the MATE's contig
        ("weird", 0x1, 2, 300, "60S60M", 2, sa("chr2,900,+,30M10N80M,60,0;"), 4),            # N in the SA CIGAR: the literal code raises KeyError
    ]
    meta, raw_end, raw = _records(rows)
    from sigtab_common import HostSel
    sel = HostSel(meta, raw_end, raw)
    lit = lambda k: tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, k), 5, "SA", names[int(sel.tid[k])])
    with pytest.raises(KeyError):
        t.add(meta, raw_end, raw, 5, literal=lit)
    t.close()
    t = SignalTables(names, [ln for _, ln in CONTIGS], 1000)
    meta, raw_end, raw = _records(rows[:-1])
    t.add(meta, raw_end, raw, 5)
    disc, seg = t.text(0)
    assert disc.decode().splitlines() == [
        "pairA\tchr1\tchr1\t101\t221\tFalse\t901\t1021\tTrue",
        "triple\tchr1\tchr1\t1001\t1121\tFalse\t1101\t1221\tTrue",
        "pairB\tchr1\tchr2\t201\t321\tTrue\t251\t371\tFalse"]
    assert seg[:, :2].tolist() == [[0, 0], [0, 2]]
    split, _ = t.text(1)
    # chr1 < chr10 as strings: chrA = chr1 for both reads of splitA; the second row's fields follow the first's on one line
    assert split.decode().splitlines() == ["splitA\tchr1\tchr10\t501\tFalse\t7060\tFalse\t501\t561\t7000\t7060\t501\tTrue\t7001\tFalse\t501\t561\t7001\t7061"]
    assert t.clips(0) == b">clipA|chr1|401\n" + b"ACGT" * 30 + b"\n"
    # the signal table: both contigs >= min_contig; find_discordant_pos by orientation; splits behind the discordants of their bucket
    n, nb = t.cluster_table(False, 1000)
    posA, posB = np.zeros(n, np.int32), np.zeros(n, np.int32)
    off, a, b = t.cluster_columns(posA, posB, nb)
    assert (n, nb) == (4, 3) and off.tolist() == [0, 2, 3, 4] and list(zip(a.tolist(), b.tolist())) == [(0, 0), (0, 1), (0, 2)]
    assert posA.tolist() == [221, 1121, 501, 201] and posB.tolist() == [901, 1101, 7060, 371]
    n, nb = t.cluster_table(True, 1000)
    t.cluster_columns(posA, posB, nb)
    assert posA.tolist() == [101, 1001, 501, 321] and posB.tolist() == [1021, 1221, 7060, 251]
    # a split row whose SA contig is not in the header: dropped when that name sorts first (chrA not in splits), KeyError when it is chrB
    for sa_chr, err in (("aaa", False), ("zzz", True)):
        u = SignalTables(names, [ln for _, ln in CONTIGS], 1000)
        m2 = _records([("q", 0x1, 0, 500, "60S60M", 0, sa("%s,7000,+,60M60S,60,0;" % sa_chr), 4)])
        if err:
            with pytest.raises(KeyError, match="zzz"):
                u.add(*m2, 5)
        else:
            u.add(*m2, 5)
            assert u.text(1)[0] == b"" and u.rows()[1]["chr1"] == [["aaa", "chr1", "q", 7060, False, 501, False, 7000, 7060, 501, 561]]
        u.close()
    t.close()


def test_unusual_sa_tags_go_through_the_literal_code_in_file_order():
    """an SA tag the C parser does not take ('+500' as position) stops the batch at that read; the caller's literal row enters the
    tables at its place, so the fragment's fields are concatenated in file order"""
    names = [n for n, _ in CONTIGS]
    sa = lambda s: [("SA", "Z", s)]
    rows = [("f", 0x1, 0, 100, "60S60M", 0, sa("chr1,7000,+,60M60S,60,0;"), 4),
            ("f", 0x1, 0, 200, "60S60M", 0, sa("chr1,+8000,+,60M60S,60,0;"), 4),          # int('+8000') == 8000 in the literal code
            ("f", 0x1, 0, 300, "60S60M", 0, sa("chr1,9000,+,60M60S,60,0;"), 4)]
    meta, raw_end, raw = _records(rows)
    from sigtab_common import HostSel
    sel = HostSel(meta, raw_end, raw)
    calls = []

    def lit(k):
        calls.append(k)
        return tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, k), 5, "SA", names[int(sel.tid[k])])
    t = SignalTables(names, [ln for _, ln in CONTIGS], 1000)
    t.add(meta, raw_end, raw, 5, literal=lit)
    assert calls == [1]
    want = []
    for k in range(3):
        want += tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, k), 5, "SA", "chr1")[3:]
    assert t.text(1)[0].decode() == "f\tchr1\tchr1\t" + "\t".join(map(str, want)) + "\n"
    assert [r[3] for r in t.rows()[1]["chr1"]] == [want[0], want[8], want[16]]
    t.close()


def test_random_rows_through_the_blob_equal_the_literal_merge(tmp_path):
    """made-up rows (fragments with one, two and three reads, mates on other contigs and on the contig below --min_contig, positions
    beyond the contig end, split fragments seen on both mates) imported as a row blob: tables == literal merge, text path == table path"""
    names = [n for n, _ in CONTIGS]
    header = {"SQ": [{"SN": n, "LN": ln} for n, ln in CONTIGS]}
    for seed in (1, 2, 3):
        rng = random.Random(seed)
        reads = []
        for f in range(500):
            q = "frag%05d" % f
            ta = rng.randrange(len(CONTIGS))
            tb = ta if rng.random() < 0.7 else rng.randrange(len(CONTIGS))
            for k in range(rng.choice([1, 2, 2, 2, 3])):
                t, mate = (ta, tb) if k % 2 == 0 else (tb, ta)
                pos = rng.randrange(0, CONTIGS[t][1] + 5000)
                reads.append((t, pos, "d", (t, mate, pos + 1, pos + 151, rng.random() < 0.5, q)))
            if rng.random() < 0.4:
                for k in range(rng.choice([1, 1, 2])):
                    t, o = (ta, tb) if k == 0 else (tb, ta)
                    a, b = (o, t) if names[o] < names[t] else (t, o)
                    pos = rng.randrange(0, CONTIGS[t][1] + 3000)
                    reads.append((t, pos, "s", (t, a, b, rng.random() < 0.5, rng.random() < 0.5, q, [pos + 1, rng.randrange(1, 60000), pos, pos + 80, pos + 5000, pos + 5070])))
        reads.sort(key=lambda r: (r[0], r[1]))
        reads = [r for r in reads if CONTIGS[r[0]][1] >= 1000]                        # worker() never runs on the small contig
        drows = [r[3] for r in reads if r[2] == "d"]
        srows = [r[3] for r in reads if r[2] == "s"]
        nm = bytearray()
        d = np.zeros(len(drows), dtype=D_ROW)
        for i, (t, mate, s, e, rev, q) in enumerate(drows):
            d[i] = (t, mate, s, e, len(nm), len(q), rev, 0, 0)
            nm += q.encode()
        s_ = np.zeros(len(srows), dtype=S_ROW)
        for i, (t, a, b, rev, sam, q, f) in enumerate(srows):
            s_[i] = (t, a, b, len(q), rev, sam, len(nm), 0, 0, 0, f)
            nm += q.encode()
        blob = np.concatenate([np.array([0x3142415447495354, len(d), len(s_), len(nm)], dtype="<u8").view(np.uint8), d.view(np.uint8), s_.view(np.uint8),
                               np.frombuffer(bytes(nm), dtype=np.uint8)])
        t = SignalTables(names, [ln for _, ln in CONTIGS], 1000)
        t.import_rows(blob)
        u = SignalTables(names, [ln for _, ln in CONTIGS], 1000)           # export -> import is the identity on what the tables say
        u.import_rows(t.export_rows())
        assert u.text(0)[0] == t.text(0)[0] and u.text(1)[0] == t.text(1)[0] and u.rows() == t.rows()
        u.close()
        ld, ls, _ = literal_files(t, header, str(tmp_path / ("r%d" % seed)), "S")
        assert t.text(0)[0] == ld and t.text(1)[0] == ls and ld.count(b"\n") > 50 and ls.count(b"\n") > 20
        assert b"scaffoldA\t" not in b"\n".join(l.split(b"\t", 2)[1] for l in ld.splitlines())      # never a chrA
        # the signal table from the tables == the one parsed from the text (tiddit_cluster._read_signals), the clip quirk included
        prefix = str(tmp_path / ("r%d" % seed))
        for is_mp in (False, True):
            sig, pos = tiddit_cluster._read_signals(prefix, ["S"], dict(CONTIGS), is_mp, 1000, True)
            n, nb = t.cluster_table(is_mp, 1000)
            posA, posB = np.zeros(n, np.int32), np.zeros(n, np.int32)
            off, a, b = t.cluster_columns(posA, posB, nb)
            order = [(x, y) for x in names if x in pos for y in names if y in pos[x]]
            assert [(names[x], names[y]) for x, y in zip(a.tolist(), b.tolist())] == order
            for k, (x, y) in enumerate(order):
                flat = [int(v) for v in pos[x][y]]
                assert posA[off[k]:off[k + 1]].tolist() == flat[0::3] and posB[off[k]:off[k + 1]].tolist() == flat[1::3], (seed, is_mp, x, y)
            assert "scaffoldA" not in pos and all("scaffoldA" not in pos[x] for x in pos)
            clipped = sum(1 for x in pos for y in pos[x] for v, ln in zip(pos[x][y][0::3], [dict(CONTIGS)[x]] * len(pos[x][y][0::3])) if int(v) == ln)
        assert clipped > 0
        t.close()


def test_candidate_dictionaries_built_with_the_cpython_api_equal_the_python_loop():
    """tiddit_amd/_pycand (csrc/tdt_pycand.c) against the Python loop of tiddit_cluster._native_candidates on random member arrays: equal
    dictionaries, the same key order at every level, the same value types, empty candidates, names with non-ASCII characters — and the
    argument checks (a row that points outside the arrays is refused, not read)"""
    _pycand = pytest.importorskip("tiddit_amd._pycand")          # (built by tiddit_amd.build when the interpreter's headers are there)
    rng = np.random.default_rng(20260930)
    nc = 400
    nd, ns = rng.integers(0, 9, nc), rng.integers(0, 5, nc)
    m = int((nd + ns).sum())
    cand = np.stack([rng.integers(0, 7, nc), rng.permutation(nc) + 3, nd, ns], 1).astype(np.int32)
    cols = [rng.integers(-5, 2 ** 31 - 1, m).astype(np.int32) for _ in range(6)]
    ori = [rng.integers(0, 2, m).astype(np.uint8) for _ in range(2)]
    frag = ["frag:%d/é%d" % (i % 97, i) for i in range(m)]          # (repeated prefixes; multi-byte characters)
    names = "\n".join(frag).encode()
    sample = "WGS"
    W = ("False", "True")
    sA, eA, sB, eB, pA, pB = (c.tolist() for c in cols)
    oA, oB = [W[x] for x in ori[0].tolist()], [W[x] for x in ori[1].tolist()]
    want = [{} for _ in range(7)]
    lo = 0
    for bkt, cid, a, b in cand.tolist():
        mid, hi = lo + a, lo + a + b
        c = tiddit_cluster._new_candidate()
        c["samples"].add(sample)
        c["sample_discordants"][sample], c["sample_splits"][sample], c["sample_contigs"][sample] = set(frag[lo:mid]), set(frag[mid:hi]), set()
        c["discordants"], c["splits"] = set(frag[lo:mid]), set(frag[mid:hi])
        for side, pos, o, s, e in (("positions_A", pA, oA, sA, eA), ("positions_B", pB, oB, sB, eB)):
            c[side]["splits"], c[side]["discordants"] = pos[mid:hi], pos[lo:mid]
            c[side]["orientation_splits"], c[side]["orientation_discordants"] = o[mid:hi], o[lo:mid]
            c[side]["start"], c[side]["end"] = s[lo:hi], e[lo:hi]
        want[bkt][cid] = c
        lo = hi
    got = [{} for _ in range(7)]
    assert _pycand.build(got, cand, names, *cols, *ori, sample) == nc
    assert got == want
    for g, w in zip(got, want):
        assert list(g) == list(w)
        for cid in g:
            assert list(g[cid]) == list(w[cid]) == list(tiddit_cluster._new_candidate())
            for side in ("positions_A", "positions_B"):
                assert list(g[cid][side]) == list(w[cid][side])
                for k, v in g[cid][side].items():
                    assert [type(x) for x in v] == [type(x) for x in w[cid][side][k]]
            assert g[cid]["discordants"] is not g[cid]["sample_discordants"][sample] and type(cid) is int
    # ... and with is_mp / min_reads given, what tiddit_cluster._finish_candidates adds, for both library kinds: counts of DISTINCT names,
    # the mode with CPython's first-inserted tie rule, the orientation vote, the extreme positions, the four region keys in their order.
    # (Positions from a small range, so that modes have ties and repeats; candidates need a member.)
    keep = (nd + ns) > 0
    small = [rng.integers(0, 6, m).astype(np.int32) for _ in range(6)]
    frag2 = ["f%d" % (i % 11) for i in range(m)]                 # repeated names inside a candidate: N_* < members
    names2 = "\n".join(frag2).encode()
    for is_mp in (False, True):
        for min_reads in (1, 3):
            ref = [{} for _ in range(7)]
            _pycand.build(ref, cand[keep], names2, *small, *ori, sample)
            wrapped = {"a": {str(k): ref[k] for k in range(7)}}
            tiddit_cluster._finish_candidates(wrapped, is_mp, min_reads)
            fin = [{} for _ in range(7)]
            _pycand.build(fin, cand[keep], names2, *small, *ori, sample, is_mp, min_reads)
            assert fin == ref
            for g, w in zip(fin, ref):
                for cid in g:
                    assert list(g[cid]) == list(w[cid]) and all(type(g[cid][k]) is type(w[cid][k]) for k in g[cid])
    with pytest.raises(ValueError):                              # a candidate without members cannot be finished (min() of nothing in the Python)
        _pycand.build([{} for _ in range(7)], np.array([[0, 1, 0, 0]], dtype=np.int32), b"", *[c[:0] for c in cols], *[o[:0] for o in ori], sample, False, 1)
    bad = cand.copy()
    bad[-1, 2] += 1                                              # one member more than the arrays hold
    with pytest.raises(ValueError):
        _pycand.build([{} for _ in range(7)], bad, names, *cols, *ori, sample)
    with pytest.raises(ValueError):
        _pycand.build([{} for _ in range(3)], cand, names, *cols, *ori, sample)         # a bucket index beyond the slots
    with pytest.raises(ValueError):
        _pycand.build([{} for _ in range(7)], cand, names, cols[0][:-1], *cols[1:], *ori, sample)
