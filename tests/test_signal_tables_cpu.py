"""Host-only checks of the signal tables' two ways through tiddit_signal.main's merge (tiddit_signal.pyx:246-326) and of the two ways
into tiddit_cluster.main's signal table (tiddit_cluster.pyx:46-137): merged while the file is scanned (EarlyTables) vs merged after it;
rows handed over in memory (written_tables) vs parsed from the .tab text.  No device is involved: the rows are made up."""
import os
import random

import pytest

from tiddit_amd import tiddit_cluster, tiddit_signal

CONTIGS = [("chr1", 50000), ("chr10", 40000), ("chr2", 30000), ("scaffoldA", 900), ("chrM", 16000)]
NAMES = [n for n, _ in CONTIGS]
MIN_CONTIG = 1000
HEADER = {"SQ": [{"SN": n, "LN": ln} for n, ln in CONTIGS]}


def make_rows(seed, n_frag=400):
    """per-contig discordant / split rows in FILE order, as scan_signals returns them (worker, :214-221 and SA_analysis :138-142):
    fragments with one, two and three reads, mates on other contigs (string order of the names decides chrA: chr10 < chr2),
    reads on the contig below --min_contig, positions beyond the contig end, split fragments seen on both mates"""
    rng = random.Random(seed)
    reads = []                                              # (tid, pos, kind, row)
    for f in range(n_frag):
        q = "frag%05d" % f
        ta = rng.randrange(len(CONTIGS))
        tb = ta if rng.random() < 0.7 else rng.randrange(len(CONTIGS))
        for k in range(rng.choice([1, 2, 2, 2, 3])):
            t = ta if k % 2 == 0 else tb
            mate = tb if k % 2 == 0 else ta
            chrom, mname = NAMES[t], NAMES[mate]
            chrA, chrB = (mname, chrom) if mname < chrom else (chrom, mname)
            pos = rng.randrange(0, CONTIGS[t][1] + 300)     # some beyond the contig's length (the clip quirk of :67-70)
            reads.append((t, pos, "d", [chrA, chrB, q, pos + 1, pos + 151, rng.random() < 0.5, chrom]))
        if rng.random() < 0.4:
            for k in range(rng.choice([1, 1, 2])):
                t = ta if k == 0 else tb
                chrom, other = NAMES[t], NAMES[tb if k == 0 else ta]
                chrA, chrB = (other, chrom) if other < chrom else (chrom, other)
                pos = rng.randrange(0, CONTIGS[t][1])
                reads.append((t, pos, "s", [chrA, chrB, q, pos + 1, rng.random() < 0.5, rng.randrange(1, 60000), rng.random() < 0.5,
                                            pos, pos + 80, pos + 5000, pos + 5070]))
    reads.sort(key=lambda r: (r[0], r[1]))                  # coordinate sorted, contigs in header order
    data = {n: [] for n in NAMES}
    splits = {n: [] for n in NAMES}
    for t, pos, kind, row in reads:
        (data if kind == "d" else splits)[NAMES[t]].append(row)
    return data, splits


def write_tables(tmp_path, tag, data, splits, early):
    chromosomes = [n for n, ln in CONTIGS if ln >= MIN_CONTIG]
    prefix = str(tmp_path / tag)
    os.makedirs(prefix + "_tiddit", exist_ok=True)
    clips = {n: [[">c|%s|1\n" % n, "ACGT\n"]] for n in NAMES}
    tiddit_signal.PREMERGED.clear()
    if early:
        tabs = tiddit_signal.EarlyTables(NAMES, chromosomes)
        batch = 37                                          # rows arrive batch by batch, contig runs inside a batch in file order
        for n in NAMES:
            for lo in range(0, max(len(data[n]), len(splits[n])), batch):
                tabs.add(n, splits[n][lo:lo + batch], "s")
                tabs.add(n, data[n][lo:lo + batch], "d")
        assert tabs.ok
        tiddit_signal.PREMERGED["tables"] = (tabs.data, tabs.splits, data, splits, tabs.slines)
    tiddit_signal._merge_and_write(HEADER, chromosomes, data, splits, clips, prefix, "S")
    d, s = prefix + "_tiddit/discordants_S.tab", prefix + "_tiddit/splits_S.tab"
    return prefix, open(d).read(), open(s).read(), open(prefix + "_tiddit/clips_S.fa").read()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tables_merged_while_scanning_equal_tables_merged_afterwards(tmp_path, seed):
    data, splits = make_rows(seed)
    _, d0, s0, c0 = write_tables(tmp_path, "late", data, splits, early=False)
    _, d1, s1, c1 = write_tables(tmp_path, "early", data, splits, early=True)
    assert d0 == d1 and s0 == s1 and c0 == c1
    assert d0.count("\n") > 50 and s0.count("\n") > 20
    # the late merge is the literal loop of :262-326; pin one property of it here: a fragment with one read is not written
    singles = {r[2] for n in NAMES for r in data[n]}
    written = {l.split("\t")[0] for l in d0.splitlines()}
    assert written < singles


def test_early_tables_stand_down_on_a_row_out_of_contig_order():
    tabs = tiddit_signal.EarlyTables(NAMES, ["chr1", "chr10", "chr2", "chrM"])
    tabs.add("chr10", [["chr10", "chr10", "q", 1, 2, True, "chr10"]], "d")
    assert tabs.ok
    tabs.add("chr1", [["chr1", "chr1", "p", 1, 2, True, "chr1"]], "d")
    assert not tabs.ok
    tabs.add("scaffoldA", [["chr1", "scaffoldA", "p", 1, 2, True, "scaffoldA"]], "d")       # (a contig main() does not walk: ignored)
    assert not tabs.ok


@pytest.mark.parametrize("early", [False, True])
@pytest.mark.parametrize("is_mp", [False, True])
def test_rows_handed_over_in_memory_equal_rows_parsed_from_the_text(tmp_path, early, is_mp):
    data, splits = make_rows(7, 600)
    prefix, d, s, _ = write_tables(tmp_path, "t", data, splits, early=early)
    lengths = dict(CONTIGS)
    paths = (prefix + "_tiddit/discordants_S.tab", prefix + "_tiddit/splits_S.tab")
    assert tiddit_signal.written_tables(*paths) is not None
    sig_m, pos_m = tiddit_cluster._read_signals(prefix, ["S"], lengths, is_mp, MIN_CONTIG, True)
    tiddit_signal.WRITTEN_TABLES.clear()
    sig_t, pos_t = tiddit_cluster._read_signals(prefix, ["S"], lengths, is_mp, MIN_CONTIG, True)
    assert [(a, list(sig_m[a])) for a in sig_m] == [(a, list(sig_t[a])) for a in sig_t]          # the same buckets in the same order
    n = 0
    for a in sig_t:
        for b in sig_t[a]:
            assert [[str(v) for v in r] for r in sig_m[a][b]] == [[str(v) for v in r] for r in sig_t[a][b]]
            assert [int(v) for v in pos_m[a][b]] == [int(v) for v in pos_t[a][b]]
            n += len(sig_t[a][b])
    assert n > 100
    assert "scaffoldA" not in sig_t and all("scaffoldA" not in sig_t[a] for a in sig_t)


def test_written_tables_are_forgotten_when_a_file_changes(tmp_path):
    data, splits = make_rows(9, 50)
    prefix, d, s, _ = write_tables(tmp_path, "w", data, splits, early=False)
    paths = (prefix + "_tiddit/discordants_S.tab", prefix + "_tiddit/splits_S.tab")
    assert tiddit_signal.written_tables(*paths) is not None
    assert tiddit_signal.written_tables(paths[0], paths[1] + ".other") is None
    with open(paths[1], "a") as f:                                      # somebody edits the splits file: its rows come from the text again
        f.write("extra\tchr1\tchr1\t10\tTrue\t20\tFalse\t1\t2\t3\t4\n")
    assert tiddit_signal.written_tables(*paths) is None
    sig, pos = tiddit_cluster._read_signals(prefix, ["S"], dict(CONTIGS), False, MIN_CONTIG, True)
    assert any(r[0] == "extra" for r in sig["chr1"]["chr1"])


def test_quiet_gc_leaves_the_collector_as_it_found_it_and_only_the_cli_freezes():
    import gc
    from tiddit_amd.hostutil import quiet_gc, thaw
    assert gc.isenabled()
    with quiet_gc():
        assert not gc.isenabled()
        with quiet_gc():                      # nested (tiddit_signal.main inside the command line's own): stays off, nothing frozen
            assert not gc.isenabled()
        assert not gc.isenabled()
    assert gc.isenabled() and gc.get_freeze_count() == 0
    with quiet_gc(freeze=True):
        pass
    assert gc.isenabled() and gc.get_freeze_count() > 0
    with quiet_gc():                          # the next stage hands the frozen objects back
        assert gc.get_freeze_count() == 0
    with quiet_gc(freeze=True):
        pass
    thaw()
    assert gc.get_freeze_count() == 0
    gc.disable()
    try:
        with quiet_gc(freeze=True):           # a host that runs without the collector keeps running without it
            pass
        assert not gc.isenabled() and gc.get_freeze_count() == 0
    finally:
        gc.enable()
