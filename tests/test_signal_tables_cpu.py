"""Host-only checks of the hand-over between tiddit_signal.main and tiddit_cluster.main of one process: the native signal tables
stay with the files they were written to (tiddit_signal.written_tables) and are dropped as soon as a file changes — then, and for
files from elsewhere, tiddit_cluster parses the text (tiddit_cluster.pyx:46-137).  (The tables themselves: tests/test_sigtab_cpu.py.)"""
import numpy as np

from tiddit_amd import tiddit_cluster, tiddit_signal
from tiddit_amd.sigtab import D_ROW, SignalTables

CONTIGS = [("chr1", 50000), ("chr10", 40000), ("chr2", 30000), ("scaffoldA", 900), ("chrM", 16000)]
NAMES = [n for n, _ in CONTIGS]
MIN_CONTIG = 1000


def small_tables():
    """three discordant fragments on chr1 (two reads each) as a row blob"""
    nm = bytearray()
    d = np.zeros(6, dtype=D_ROW)
    for i in range(6):
        q = "frag%d" % (i // 2)
        d[i] = (0, 0, 1000 + 10 * i, 1150 + 10 * i, len(nm), len(q), i % 2, 0, 0)
        nm += q.encode()
    blob = np.concatenate([np.array([0x3142415447495354, 6, 0, len(nm)], dtype="<u8").view(np.uint8), d.view(np.uint8), np.frombuffer(bytes(nm), dtype=np.uint8)])
    t = SignalTables(NAMES, [ln for _, ln in CONTIGS], MIN_CONTIG)
    t.import_rows(blob)
    return t


def test_written_tables_are_forgotten_when_a_file_changes(tmp_path):
    t = small_tables()
    prefix = str(tmp_path / "w")
    tiddit_signal._write_tables(t, t, [n for n, ln in CONTIGS if ln >= MIN_CONTIG], prefix, "S")
    paths = (prefix + "_tiddit/discordants_S.tab", prefix + "_tiddit/splits_S.tab")
    assert open(paths[0]).read().count("\n") == 3 and open(paths[1]).read() == ""
    assert tiddit_signal.written_tables(*paths) is t and tiddit_signal.table_owners(*paths) is None
    assert tiddit_signal.written_tables(paths[0], paths[1] + ".other") is None
    got = tiddit_cluster._handed_over(prefix, NAMES, dict(CONTIGS), ["S"], MIN_CONTIG, True)[0]
    assert got is t
    # other contigs, another sample list, assembly contigs: the text decides
    assert tiddit_cluster._handed_over(prefix, NAMES[:-1], dict(CONTIGS), ["S"], MIN_CONTIG, True)[0] is None
    assert tiddit_cluster._handed_over(prefix, NAMES, dict(CONTIGS, chr2=31000), ["S"], MIN_CONTIG, True)[0] is None
    assert tiddit_cluster._handed_over(prefix, NAMES, dict(CONTIGS), ["S", "T"], MIN_CONTIG, True)[0] is None
    assert tiddit_cluster._handed_over(prefix, NAMES, dict(CONTIGS), ["S"], MIN_CONTIG, False)[0] is None
    with open(paths[1], "a") as f:                                      # somebody edits the splits file: its rows come from the text again
        f.write("extra\tchr1\tchr1\t10\tTrue\t20\tFalse\t1\t2\t3\t4\n")
    assert tiddit_signal.written_tables(*paths) is None
    sig, pos = tiddit_cluster._read_signals(prefix, ["S"], dict(CONTIGS), False, MIN_CONTIG, True)
    assert any(r[0] == "extra" for r in sig["chr1"]["chr1"]) and len(sig["chr1"]["chr1"]) == 4
    # the next main() of the process replaces (and frees) the tables it kept
    u = small_tables()
    tiddit_signal._write_tables(u, u, [n for n, ln in CONTIGS if ln >= MIN_CONTIG], prefix, "S")
    assert t._h is None and tiddit_signal.written_tables(*paths) is u
    tiddit_signal._forget_tables()
    assert u._h is None


def test_background_writes_hand_the_tables_over_at_once_and_finish_with_the_same_files(tmp_path):
    """what the one-process command line does (tiddit_signal.BACKGROUND_WRITES): the blocks are placed by a thread, the tables are handed
    to tiddit_cluster while it runs, finish_writes() leaves the files of the synchronous writer and an entry with stamps"""
    kept = [n for n, ln in CONTIGS if ln >= MIN_CONTIG]
    t = small_tables()
    t.add_clips(0, b">clipA|chr1|1000\nACGT\n")
    ref_prefix = str(tmp_path / "sync")
    tiddit_signal._write_tables(t, t, kept, ref_prefix, "S")
    want = {k: open(ref_prefix + "_tiddit/" + k, "rb").read() for k in ("discordants_S.tab", "splits_S.tab", "clips_S.fa", "clips/chr1.fa")}
    assert want["clips_S.fa"] == want["clips/chr1.fa"] and want["clips_S.fa"]
    u = small_tables()
    u.add_clips(0, b">clipA|chr1|1000\nACGT\n")
    prefix = str(tmp_path / "bg")
    paths = (prefix + "_tiddit/discordants_S.tab", prefix + "_tiddit/splits_S.tab")
    tiddit_signal.BACKGROUND_WRITES = True
    try:
        tiddit_signal._write_tables(u, u, kept, prefix, "S")
    finally:
        tiddit_signal.BACKGROUND_WRITES = False
    assert tiddit_signal.written_tables(*paths) is u                    # handed over while (or after) the thread runs, without a stamp
    assert tiddit_cluster._handed_over(prefix, NAMES, dict(CONTIGS), ["S"], MIN_CONTIG, True)[0] is u
    tiddit_signal.finish_writes()
    assert set(tiddit_signal.WRITE_SECONDS) == {"writer thread", "waited for it"}
    got = {k: open(prefix + "_tiddit/" + k, "rb").read() for k in want}
    assert got == want
    assert tiddit_signal.written_tables(*paths) is u and tiddit_signal.WRITTEN_TABLES[tuple(__import__("os").path.abspath(p) for p in paths)][0] is not None
    with open(paths[0], "a") as f:                                      # stamped now: an edit drops the hand-over as ever
        f.write("x\n")
    assert tiddit_signal.written_tables(*paths) is None
    tiddit_signal.finish_writes()                                       # nothing pending: a no-op
    assert tiddit_signal.WRITE_SECONDS == {}
    tiddit_signal._forget_tables()
    assert u._h is None


def test_a_declined_hand_over_waits_for_the_writer_thread_before_the_text_is_parsed(tmp_path, monkeypatch):
    """BACKGROUND_WRITES: the files exist at their final size while a thread is still placing their blocks.  When tiddit_cluster.main of
    the same process does NOT take the tables over (here: another contig length than the tables were made for) it parses the text — it
    must first wait for that thread, or it reads zero-filled / partial lines"""
    import threading
    import time
    import oracle
    kept = [n for n, ln in CONTIGS if ln >= MIN_CONTIG]

    def buckets_by_oracle(buckets, epsilon, m, **kw):
        out = []
        for b in buckets:
            order = np.argsort(b[:, 0], kind="stable")
            lab = np.empty(len(b))
            lab[order] = oracle.dbscan_main(b[order], epsilon, m)
            out.append(lab)
        return out
    monkeypatch.setattr(tiddit_cluster, "cluster_buckets", buckets_by_oracle)
    other = dict(CONTIGS, chr2=31000)                                  # declines the hand-over (tiddit_cluster._handed_over)
    args = (NAMES, other, ["S"], False, 100, 2, 1000, MIN_CONTIG, True, 2)
    t = small_tables()
    ref_prefix = str(tmp_path / "sync")
    tiddit_signal._write_tables(t, t, kept, ref_prefix, "S")
    want = tiddit_cluster.main(ref_prefix, *args)
    assert "parse .tab" in tiddit_cluster.STAGE_SECONDS and want["chr1"]["chr1"]
    u = small_tables()
    started = threading.Event()
    real = u.pwrite

    def slow_pwrite(*a):
        started.set()
        time.sleep(0.3)
        return real(*a)
    u.pwrite = slow_pwrite
    prefix = str(tmp_path / "bg")
    tiddit_signal.BACKGROUND_WRITES = True
    try:
        tiddit_signal._write_tables(u, u, kept, prefix, "S")
    finally:
        tiddit_signal.BACKGROUND_WRITES = False
    assert started.wait(5)
    ent = next(iter(tiddit_signal.WRITTEN_TABLES.values()))
    assert ent[4] is not None and ent[4]["thread"].is_alive()          # the blocks are not placed yet
    got = tiddit_cluster.main(prefix, *args)
    assert "parse .tab" in tiddit_cluster.STAGE_SECONDS
    assert not ent[4]["thread"].is_alive() and got == want
    tiddit_signal._forget_tables()


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


def test_selected_reads_make_their_per_record_views_on_demand():
    """SelectedReads (what the device scan returns per batch): raw_bytes / rec_off / sa_off are only needed by RecordView; they are made
    when first asked for, once, and say what the eager code said"""
    meta = np.zeros(3, dtype=tiddit_signal._META)
    meta["tid"], meta["pos"], meta["sa_rel"] = [0, 0, 1], [10, 20, 30], [-1, 40, 7]
    raw_end = np.array([100, 250, 300], dtype=np.uint32)
    raw = (np.arange(300) % 251).astype(np.uint8)
    sel = tiddit_signal.SelectedReads(None, meta, raw_end, raw)
    assert not any(k in sel.__dict__ for k in ("_raw_bytes", "_rec_off", "_sa_off")) and len(sel) == 3
    assert sel.rec_off.tolist() == [0, 100, 250] and sel.rec_off.dtype == np.uint64
    assert sel.sa_off.tolist() == [-1, 140, 257]
    assert sel.raw_bytes == raw.tobytes() and sel.raw_bytes is sel.raw_bytes and sel.rec_off is sel.rec_off
    empty = tiddit_signal.SelectedReads(None, meta[:0], raw_end[:0], raw[:0])
    assert len(empty.rec_off) == 0 and len(empty.sa_off) == 0 and empty.raw_bytes == b""
