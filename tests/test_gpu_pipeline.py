"""GPU tests of the callers either side of the kernels: `tiddit --cov` (BASELINE configs[0]),
tiddit_signal.main and `tiddit --sv --skip_assembly` on synthetic BAMs, against the per-read literal
restatement in oracle/signal_oracle.py and the golden .bed/.wig checksums of the real reference."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from oracle import signal_oracle
from tiddit_amd import synth

pytestmark = pytest.mark.gpu

CONTIGS = [("chr1", 90000), ("chr10", 70000), ("chr2", 80000), ("chrM", 6000), ("tiny", 500)]


@pytest.fixture(scope="module")
def sv_bam(tmp_path_factory):
    from tiddit_amd import synth_bam
    d = tmp_path_factory.mktemp("sv")
    p = str(d / "SYN.bam")
    info = synth_bam.write_synthetic_bam(p, CONTIGS, depth=8, seed=11, n_events=14)
    fa = str(d / "ref.fa")
    with open(fa, "w") as f:
        for i, (n, l) in enumerate(CONTIGS):
            s = synth.gen_sequence(l, seed=100 + i).tobytes().decode()
            f.write(">%s\n" % n + "\n".join(s[k:k + 70] for k in range(0, l, 70)) + "\n")
    return p, fa, info, d


def test_cli_cov_config1_bed_and_wig_sha(golden_dir, tmp_path):
    """BASELINE configs[0]: `tiddit --cov` on the 1 Mb / 10x single-contig stream, -z 500 (q 20)."""
    from tiddit_amd import __main__ as cli
    from tiddit_amd.bamio import BamWriter
    g = json.load(open(os.path.join(golden_dir, "coverage.json")))["config1"]["cov"]
    start, end, mapq, flag = synth.gen_reads(1_000_000, 10)
    bam = str(tmp_path / "c1.bam")
    w = BamWriter(bam, [("chrS", 1_000_000)])
    for i in range(len(start)):
        ref = int(end[i] - start[i])
        cig = "%dM" % ref if ref >= 150 else "%dS%dM" % (150 - ref, ref)
        if ref > 150:
            cig = "75M%dD75M" % (ref - 150)
        w.write("r%d" % i, int(flag[i]), 0, int(start[i]), int(mapq[i]), cig, 0, int(start[i]), 0, seq="")
    w.close()
    out = str(tmp_path / "cov")
    cli.main(["--cov", "--bam", bam, "-o", out, "-z", "500"])
    assert hashlib.sha256(open(out + ".bed", "rb").read()).hexdigest() == g["bed_sha256"]
    cli.main(["--cov", "--bam", bam, "-o", out, "-z", "500", "-w"])
    assert hashlib.sha256(open(out + ".wig", "rb").read()).hexdigest() == g["wig_sha256"]


def test_cli_cov_multi_contig_vs_oracle(sv_bam, tmp_path):
    from tiddit_amd import __main__ as cli, tiddit_coverage
    bam, fa, info, d = sv_bam
    hdr, reads = signal_oracle.parse_bam(bam)
    want = signal_oracle.cov_main(hdr, reads, 500, 20)
    out = str(tmp_path / "o")
    cli.main(["--cov", "--bam", bam, "-o", out])
    ref_bed = str(tmp_path / "ref.bed")
    tiddit_coverage.print_coverage(want, hdr, 500, "bed", ref_bed)     # formatting only; values come from the oracle
    assert open(out + ".bed").read() == open(ref_bed).read()


def test_signal_main_vs_literal_restatement(sv_bam, tmp_path):
    from tiddit_amd import tiddit_signal
    bam, fa, info, d = sv_bam
    hdr, reads = signal_oracle.parse_bam(bam)
    for min_q, max_ins, min_contig in ((5, 600, 10000), (20, 450, 1000)):
        prefix = str(tmp_path / ("s%d" % min_q))
        os.makedirs(prefix + "_tiddit/clips")
        cov = tiddit_signal.main(bam, fa, prefix, min_q, max_ins, "SYN", 1, min_contig, False, 60, 25)
        wcov, wdisc, wsplit, wclips, wclip_each = signal_oracle.signal_main(hdr, reads, min_q, max_ins, "SYN", min_contig, 60, 25)
        assert list(cov) == list(wcov)
        for c in wcov:
            assert np.array_equal(cov[c], wcov[c]), c
        assert open(prefix + "_tiddit/discordants_SYN.tab").read() == wdisc
        assert open(prefix + "_tiddit/splits_SYN.tab").read() == wsplit
        assert open(prefix + "_tiddit/clips_SYN.fa").read() == wclips
        for c, txt in wclip_each.items():
            assert open(prefix + "_tiddit/clips/%s.fa" % c).read() == txt
        assert wdisc.count("\n") > 20 and wsplit.count("\n") > 5 and wclips.count(">") > 5


def test_stats_vs_per_read_loop(sv_bam):
    from tiddit_amd import tiddit_stats
    bam, fa, info, d = sv_bam
    hdr, reads = signal_oracle.parse_bam(bam)
    for n_reads in (100000, 777):
        lib = tiddit_stats.statistics(bam, fa, 5, 100000, n_reads)
        # tiddit_stats.py:17-47, read by read
        rl, ins, innie, outtie, ns = [], [], 0, 0, 0
        for r in reads:
            if r.reference_id < 0:
                continue
            rl.append(len(r.query_sequence))
            ns += 1
            if ns > n_reads:
                break
            if r.mate_is_unmapped or r.is_reverse == bool(r.flag & 0x20):
                continue
            if r.next_reference_name != r.reference_name or r.isize > 100000:
                continue
            if r.mate_pos < r.reference_start:
                continue
            if r.is_supplementary or r.is_secondary or r.is_duplicate or r.mapq < 5:
                continue
            ins.append(r.isize)
            if r.is_reverse and not (r.flag & 0x20):
                outtie += 1
            else:
                innie += 1
        assert lib["avg_read_length"] == np.average(rl)
        assert lib["avg_insert_size"] == np.average(ins) and lib["std_insert_size"] == np.std(ins)
        assert lib["percentile_insert_size"] == np.percentile(ins, 99.9)
        assert lib["mp"] == (not innie > outtie)


def test_cli_sv_skip_assembly_end_to_end(sv_bam, tmp_path):
    from tiddit_amd import __main__ as cli
    bam, fa, info, d = sv_bam
    out = str(tmp_path / "svrun")
    cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--min_contig", "5000"])
    for f in ("_tiddit/discordants_SYN.tab", "_tiddit/splits_SYN.tab", "_tiddit/clips_SYN.fa", ".ploidies.tab", ".candidates.tab"):
        assert os.path.isfile(out + f), f
    rows = [l.split("\t") for l in open(out + ".candidates.tab") if not l.startswith("#")]
    assert len(rows) >= 5
    found = 0
    for ev in info["events"]:
        if ev["type"] == "DEL":
            for r in rows:
                if r[0] == ev["chrom"] and r[2] == ev["chrom"] and abs(int(r[1]) - ev["start"]) < 400 and abs(int(r[3]) - ev["end"]) < 400:
                    found += 1
                    break
    n_del = sum(1 for ev in info["events"] if ev["type"] == "DEL")
    assert found >= n_del - 1, (found, n_del)
    plo = open(out + ".ploidies.tab").read().splitlines()
    assert plo[0] == "Chromosome\tPloidy\tPloidy_rounded\tMean_coverage" and len(plo) >= 4


def test_cli_sv_error_behind_the_scan_leaves_no_helper_thread(sv_bam, tmp_path, monkeypatch):
    """the one-process `tiddit --sv` places its signal files from a writer thread beside the ploidy table and the clustering: when one of
    those stages raises, the error reaches the caller, the writer thread has been joined and the files it was placing are complete"""
    import threading
    from tiddit_amd import __main__ as cli, tiddit_cluster
    bam, fa, info, d = sv_bam
    out = str(tmp_path / "ok")
    cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--min_contig", "5000"])
    want = {k: open(out + "_tiddit/" + k, "rb").read() for k in ("discordants_SYN.tab", "splits_SYN.tab", "clips_SYN.fa")}

    def boom(*a, **k):
        raise RuntimeError("clustering failed")
    monkeypatch.setattr(tiddit_cluster, "main", boom)
    out2 = str(tmp_path / "bad")
    with pytest.raises(RuntimeError, match="clustering failed"):
        cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out2, "--skip_assembly", "--min_contig", "5000"])
    assert not [t for t in threading.enumerate() if t.name in ("tiddit-signal-writer", "tiddit-gc")]
    assert {k: open(out2 + "_tiddit/" + k, "rb").read() for k in want} == want


def test_region_counts_vs_literal_loop(sv_bam):
    """tiddit_variant.get_region's loop (row §8(f)3): one wavefront per candidate vs the per-read restatement"""
    from tiddit_amd import tiddit_region
    bam, fa, info, d = sv_bam
    table = tiddit_region.ReadTable(bam)
    rng = np.random.default_rng(4)
    for chrom, LN in CONTIGS[:4]:
        t = table.tid[chrom]
        nq = 300
        starts = rng.integers(0, LN - 10, nq)
        ends = np.minimum(starts + rng.integers(0, 3000, nq), LN + 500)        # some regions run past the contig end
        bps = np.where(rng.random(nq) < 0.5, starts, ends)
        for ev in info["events"]:                                                # ... and the planted breakpoints
            if ev.get("chrom") == chrom:
                starts[0], ends[0], bps[0] = ev["start"] - 100, ev["start"] + 100, ev["start"]
        for min_q, max_ins in ((5, 600), (20, 450)):
            got = tiddit_region.region_counts(table, chrom, starts, ends, bps, min_q, max_ins)
            for q in range(nq):
                want = oracle.get_region_counts(table.contigs[t], t, LN, int(starts[q]), int(ends[q]), int(bps[q]), min_q, max_ins)
                assert np.array_equal(got[q], want), (chrom, q, starts[q], ends[q], bps[q])
        assert got[:, 1].sum() > 0 and got[:, 0].sum() > 0
    cov, flq, nd, ns, cf, cr = tiddit_region.get_region(table, "chr1", 1000, 1500, 1200, 5, 600)
    w = oracle.get_region_counts(table.contigs[0], 0, CONTIGS[0][1], 1000, 1500, 1200, 5, 600)
    assert cov == w[0] / 501 and nd == w[3] and ns == w[4] and cf == w[5] and cr == w[6]


def test_signal_worker_has_the_reference_shape(sv_bam, tmp_path):
    """tiddit_signal.worker (per-contig entry point of the reference, tiddit_signal.pyx:147-228): same rows as the whole-file scan"""
    from tiddit_amd import tiddit_signal
    bam, fa, info, d = sv_bam
    header, chroms, cov, data, splits, clips = tiddit_signal.scan_signals(bam, 5, 600, 0, 30, 20, 50)
    prefix = str(tmp_path / "w")
    for chrom in chroms[:3]:
        name, rows, srows, bins, path = tiddit_signal.worker(chrom, bam, fa, prefix, 5, 600, "SYN", 50, True, 30, 20)
        assert name == chrom and rows == data[chrom] and srows == splits[chrom] and np.array_equal(bins, cov[chrom])
        assert open(path, "rb").read() == b"".join(tiddit_signal._clip_bytes(c) for c in clips[chrom])


def test_bench_runs_every_section_on_two_ranks_sharing_the_gpu(tmp_path):
    """`bench.py --gpus 2` (the driver's multi-GPU invocation, torch.distributed.run, one process per rank) with TIDDIT_BENCH_SHARE_GPU=1:
    both ranks on the box's one GPU, exchange over gloo — every section takes its N-rank path (contigs / bases split, the shared bucket
    list with its all-gather, ONE BAM as byte-range shards with the exact all-reduce, `tiddit --sv` as one job on two ranks) and rank 0
    prints ONE JSON line with the contract's keys; the N-rank sv_e2e candidates equal the one-rank run's"""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    small = ["--steps", "2", "--warmup", "1", "--contigs", "4", "--contig-len", "8000000", "--dbscan-n", "300000", "--gc-len", "50000000",
             "--sv-mb", "3", "--ingest-mb", "1", "--no-next", "--no-cpu-baseline"]
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ, TIDDIT_BENCH_SHARE_GPU="1", TIDDIT_BENCH_TMP=str(tmp_path), TIDDIT_BENCH_DETAIL=detail)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2"] + small, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) <= 5000 and len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1      # the driver keeps a bounded tail of stdout
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["unit"] == "bins/s"
    for k in ("metric", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    for section in ("coverage_sv", "dbscan_shared", "gc", "ingest", "sv_e2e"):
        assert section in d, section
    assert set(d["roofline"]["sections"]) >= {"coverage_sv", "gc", "ingest", "sv_e2e"}
    full = json.loads(open(detail).read())                       # the detailed record of the same run
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and set(d) - {"detail"} <= set(full)
    assert "2 byte-range shards" in full["ingest"]["config"]["workload"] and "ONE job on 2 ranks" in full["sv_e2e"]["config"]["workload"]
    one = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + small + ["--no-gc", "--no-ingest", "--no-dbscan", "--no-cov-sv"],
                         env=dict(os.environ, TIDDIT_BENCH_TMP=str(tmp_path)), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["sv_e2e"]["candidates"] == d["sv_e2e"]["candidates"] > 0
    a = open(os.path.join(str(tmp_path), "tiddit_bench_sv_3", "run1.candidates.tab")).read()
    b = open(os.path.join(str(tmp_path), "tiddit_bench_sv_3", "run2.candidates.tab")).read()
    assert a == b


def test_signal_main_on_a_file_that_is_not_in_contig_order(sv_bam, tmp_path):
    """the rows are merged into the native (chrA, chrB, fragment) tables while the file is scanned — legal only while the records come in
    the header's contig order; here the records of the first contig are moved behind all others: the incremental merge must switch
    itself off and the tables must still equal the literal restatement (which visits the contigs in header order, tiddit_signal.pyx:259-284)"""
    import struct
    from tiddit_amd import bamio, tiddit_signal
    bam, fa, info, d = sv_bam
    raw = b"".join(bamio.bgzf_blocks(open(bam, "rb")))
    rd = bamio.BamReader(bam)
    skip = rd.header_bytes
    rd.close()
    first, rest, o = [], [], skip
    while o + 4 <= len(raw):
        n = 4 + struct.unpack_from("<I", raw, o)[0]
        (first if struct.unpack_from("<i", raw, o + 4)[0] == 0 else rest).append(raw[o:o + n])
        o += n
    assert first and rest
    shuffled = raw[:skip] + b"".join(rest) + b"".join(first)
    path = str(tmp_path / "moved.bam")
    with open(path, "wb") as f:
        for k in range(0, len(shuffled), 0xff00):
            f.write(bamio._bgzf_block(shuffled[k:k + 0xff00], 1))
        f.write(bamio._BGZF_EOF)
    hdr, reads = signal_oracle.parse_bam(path)
    prefix = str(tmp_path / "m")
    os.makedirs(prefix + "_tiddit/clips")
    cov = tiddit_signal.main(path, fa, prefix, 5, 600, "SYN", 1, 1000, False, 60, 25)
    tables = tiddit_signal.written_tables(prefix + "_tiddit/discordants_SYN.tab", prefix + "_tiddit/splits_SYN.tab")
    assert tables is not None and not tables.stats()["discordants_in_order"]       # the incremental merge stood down
    wcov, wdisc, wsplit, wclips, wclip_each = signal_oracle.signal_main(hdr, reads, 5, 600, "SYN", 1000, 60, 25)
    assert open(prefix + "_tiddit/discordants_SYN.tab").read() == wdisc and wdisc.count("\n") > 20
    assert open(prefix + "_tiddit/splits_SYN.tab").read() == wsplit
    assert open(prefix + "_tiddit/clips_SYN.fa").read() == wclips
    for c in wcov:
        assert np.array_equal(cov[c], wcov[c]), c
    # and on the sorted file the rows were merged while it was scanned
    p2 = str(tmp_path / "s")
    os.makedirs(p2 + "_tiddit/clips")
    tiddit_signal.main(bam, fa, p2, 5, 600, "SYN", 1, 1000, False, 60, 25)
    st = tiddit_signal.written_tables(p2 + "_tiddit/discordants_SYN.tab", p2 + "_tiddit/splits_SYN.tab").stats()
    assert st["discordants_in_order"] and st["splits_in_order"]
    tiddit_signal._forget_tables()
