"""BASELINE configs[3] on the GPU: `tiddit --sv --skip_assembly` end to end on a WGS-shaped synthetic BAM (24 chromosomes with
GRCh38's relative lengths + chrM + two scaffolds below --min_contig, 30x, 150-bp pairs, planted DEL/DUP/INV/BND with SA-tagged
split reads; tests/sv_e2e_common.py regenerates it from the fixture's seeds) at 3 Mb, 24 Mb and 240 Mb — the last one is the very file
bench.py's sv_e2e section times (48 M records, 4.3 GB).  Everything the run leaves behind is compared with tests/golden/sv_e2e*.json: signal tables and clip FASTA (restatement of tiddit_signal.pyx), 50-bp coverage of every contig,
GC bins, the ploidy table (compiled tiddit_coverage_analysis) and the ENTIRE candidates dictionary of the compiled
tiddit_cluster.main — keys, breakpoints, regions, insertion order."""
import hashlib
import os

import numpy as np
import pytest

from oracle import cluster_oracle, signal_oracle

from sv_e2e_common import load_fixture, materialise

pytestmark = pytest.mark.gpu


def h(t):
    return hashlib.sha256(t.encode()).hexdigest()


# 3 Mb, 24 Mb, 240 Mb (the bench's file), and a 25-Mb genome under a header shaped like the GRCh38 analysis set's: 3 366 contigs (alt / decoy /
# HLA names with `*` and `:`), a header of four BGZF blocks, 3 108 contigs below --min_contig, reads on 1 863 contigs, a tid = -1 tail
@pytest.fixture(scope="module", params=["sv_e2e_small.json", "sv_e2e.json", "sv_e2e_large.json", "sv_e2e_grch38.json"])
def run(request, golden_dir, tmp_path_factory):
    from tiddit_amd import __main__ as cli
    fx = load_fixture(golden_dir, request.param)
    d = str(tmp_path_factory.mktemp("e2e"))
    bam, fa, contigs = materialise(fx, d, threads=min(16, os.cpu_count() or 1))
    out = os.path.join(d, "run")
    P = fx["params"]
    cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "-s", str(P["n_reads_stats"])])
    return fx, bam, fa, contigs, out


def test_signal_files(run):
    fx, bam, fa, contigs, out = run
    assert h(open(out + "_tiddit/discordants_WGS.tab").read()) == fx["discordants_sha256"]
    assert h(open(out + "_tiddit/splits_WGS.tab").read()) == fx["splits_sha256"]
    assert h(open(out + "_tiddit/clips_WGS.fa").read()) == fx["clips_sha256"]


def test_ploidy_table(run):
    """determine_ploidy (tiddit_coverage_analysis.pyx:9-41): device GC bins + device masked medians -> the reference's table, byte for byte"""
    fx, bam, fa, contigs, out = run
    assert open(out + ".ploidies.tab").read() == fx["ploidies_tab"]


def test_candidates_whole_dictionary(run):
    from tiddit_amd import tiddit_cluster
    fx, bam, fa, contigs, out = run
    P = fx["params"]
    names = [n for n, _ in contigs]
    cand = tiddit_cluster.main(out, names, dict(contigs), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                               fx["library"]["percentile_insert_size"], P["min_contig"], True, P["min_reads"])
    assert cluster_oracle.summary(cand) == fx["candidates"]
    assert h(cluster_oracle.canonical(cand)) == fx["candidates_sha256"]
    # ... through either way in: the rows the signal stage of this process left behind (the CLI's own path), and the .tab text
    from tiddit_amd import tiddit_signal
    had = tiddit_signal.written_tables(out + "_tiddit/discordants_WGS.tab", out + "_tiddit/splits_WGS.tab") is not None
    assert ("signal table (native, into pinned columns)" in tiddit_cluster.STAGE_SECONDS) == had
    tiddit_signal._forget_tables()
    text = tiddit_cluster.main(out, names, dict(contigs), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                               fx["library"]["percentile_insert_size"], P["min_contig"], True, P["min_reads"])
    assert h(cluster_oracle.canonical(text)) == fx["candidates_sha256"] and "parse .tab" in tiddit_cluster.STAGE_SECONDS
    assert had or os.environ.get("TIDDIT_HOST_INGEST") == "1", "the signal stage's rows were not there for tiddit_cluster.main"
    # what the CLI wrote is the same table
    rows = [l.rstrip("\n").split("\t") for l in open(out + ".candidates.tab") if not l.startswith("#")]
    want = [[r[0], str(r[3]), r[1], str(r[4]), str(r[2])] + [str(x) for x in r[5:]] for r in fx["candidates"]]
    assert rows == want


def test_coverage_gc_and_library(run):
    from tiddit_amd import tiddit_gc, tiddit_signal, tiddit_stats
    fx, bam, fa, contigs, out = run
    P = fx["params"]
    lib = tiddit_stats.statistics(bam, fa, P["min_q"], 100000, P["n_reads_stats"])
    for k, v in fx["library"].items():
        assert lib[k] == v, (k, lib[k], v)
    header, chroms, cov, data, splits, clips = tiddit_signal.scan_signals(bam, P["min_q"], lib["percentile_insert_size"], P["min_contig"],
                                                                          P["min_anchor_len"], P["min_clip_len"], 50)
    assert list(cov) == list(fx["coverage_sha256"])
    for c in cov:
        assert hashlib.sha256(cov[c].astype("<f8").tobytes()).hexdigest() == fx["coverage_sha256"][c], c
    gc = tiddit_gc.main(fa, [n for n, _ in contigs], 1, 50, 0.5)
    for c, want in fx["gc_sha256"].items():
        assert hashlib.sha256(np.ascontiguousarray(gc[c]).tobytes()).hexdigest() == want, c
    if fx.get("gc_sha256_rest"):                                 # thousands of contigs: one checksum over all those below --min_contig, header order
        rest = np.concatenate([np.ascontiguousarray(gc[n]).view(np.uint8).ravel() for n, ln in contigs if ln < P["min_contig"]])
        assert hashlib.sha256(rest.tobytes()).hexdigest() == fx["gc_sha256_rest"]


def test_live_oracle_on_the_same_file(run, tmp_path):
    """the restatements run here, on this machine's copy of the file: signal tables, coverage and candidates of the product run equal them"""
    from tiddit_amd import tiddit_cluster
    fx, bam, fa, contigs, out = run
    if fx["params"]["total_mb"] > 8:
        pytest.skip("the fixture's checksums cover the large file; the live oracle pass runs on the small one")
    P = fx["params"]
    max_ins = fx["library"]["percentile_insert_size"]
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], max_ins, "WGS", P["min_contig"], P["min_anchor_len"], P["min_clip_len"])
    assert open(out + "_tiddit/discordants_WGS.tab").read() == disc
    assert open(out + "_tiddit/splits_WGS.tab").read() == split
    assert open(out + "_tiddit/clips_WGS.fa").read() == clips
    for c, txt in each.items():
        assert open(out + "_tiddit/clips/%s.fa" % c).read() == txt
    names = [n_ for n_, _ in contigs]
    args = (names, dict(contigs), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"], max_ins, P["min_contig"], True, P["min_reads"])
    assert cluster_oracle.canonical(tiddit_cluster.main(out, *args)) == cluster_oracle.canonical(cluster_oracle.main(out, *args))


# ---- BASELINE configs[4]: the same run as N processes (one per GPU in production; here they share the box's one GPU and
# exchange over gloo, like tests/test_gpu_ingest.py::test_coverage_sharded_multi_process)

def _sv_rank(rank, world, port, q, argv, min_cut):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      TIDDIT_HIP_DEVICE="0", TIDDIT_DIST_BACKEND="gloo", TIDDIT_INGEST_CHUNK=str(48 << 20), TIDDIT_CLUSTER_MIN_CUT=str(min_cut))
    try:
        import torch.distributed as dist
        from tiddit_amd import __main__ as cli, tiddit_cluster
        seen = []
        real = tiddit_cluster.cluster_columns_device

        def spy(posA, *a, **k):
            seen.append(len(posA))
            return real(posA, *a, **k)
        tiddit_cluster.cluster_columns_device = spy
        cli.main(argv)
        q.put((rank, seen))
        if dist.is_initialized():                  # (the CLI destroys the group it created)
            dist.destroy_process_group()
    except BaseException:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_sv_on_n_ranks_is_byte_identical(run, tmp_path, world):
    """`tiddit --sv --skip_assembly` as 2 / 3 ranks on ONE file: byte-range shards of the BAM with checked seams, one exact all-reduce
    of the 50-bp bins, rows sent once to the owner rank of their chrA, every owner formatting / placing its blocks of the .tab files and
    clustering / regrouping its own buckets — discordants/splits .tab, every clip FASTA, .ploidies.tab and .candidates.tab equal the
    single-process run byte for byte"""
    import socket
    import torch.multiprocessing as mp
    fx, bam, fa, contigs, out = run
    P = fx["params"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    nout = str(tmp_path / "ranks")
    argv = ["--sv", "--bam", bam, "--ref", fa, "-o", nout, "--skip_assembly", "-s", str(P["n_reads_stats"])]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sv_rank, args=(r, world, port, q, argv, 40)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(120)
    assert all(isinstance(v, list) for v in res.values()), res
    for rel in ["_tiddit/discordants_WGS.tab", "_tiddit/splits_WGS.tab", "_tiddit/clips_WGS.fa", ".ploidies.tab", ".candidates.tab"] + \
               ["_tiddit/clips/%s.fa" % n for n, ln in contigs if ln >= P["min_contig"]]:
        assert open(nout + rel, "rb").read() == open(out + rel, "rb").read(), rel
    assert h(open(nout + "_tiddit/discordants_WGS.tab").read()) == fx["discordants_sha256"]
    # every rank clustered a share of the signals (one device call each), none of them all of it
    shares = [sum(v) for v in res.values()]
    assert all(len(v) == 1 for v in res.values()) and max(shares) < sum(shares)


def test_sv_n_rank_path_over_real_rccl_with_one_rank(run, tmp_path):
    """The N-rank job's collectives over REAL RCCL: backend nccl refuses two ranks on one device, so the 2 / 3-rank runs above exchange
    over gloo and the nccl-only branches (all_to_all_single on device tensors, the asynchronous broadcast of the library statistics,
    device-side all-gathers, the all-reduce of the bins) would never execute on a one-GPU box.  TIDDIT_FORCE_DIST=1 takes the N-rank
    path with WORLD_SIZE=1: one process, one rank, every collective a real RCCL call — same files, byte for byte."""
    import socket
    import subprocess
    import sys
    fx, bam, fa, contigs, out = run
    if fx["params"]["total_mb"] > 30:
        pytest.skip("the smaller files cover it")
    P = fx["params"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    nout = str(tmp_path / "rccl1")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TIDDIT_FORCE_DIST="1",
               TIDDIT_INGEST_CHUNK=str(48 << 20))
    env.pop("TIDDIT_DIST_BACKEND", None)                       # nccl
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "tiddit_amd", "--sv", "--bam", bam, "--ref", fa, "-o", nout, "--skip_assembly", "-s", str(P["n_reads_stats"])],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for rel in ["_tiddit/discordants_WGS.tab", "_tiddit/splits_WGS.tab", "_tiddit/clips_WGS.fa", ".ploidies.tab", ".candidates.tab"] + \
               ["_tiddit/clips/%s.fa" % n for n, ln in contigs if ln >= P["min_contig"]]:
        assert open(nout + rel, "rb").read() == open(out + rel, "rb").read(), rel


def test_sv_60x_on_one_and_two_ranks(golden_dir, tmp_path):
    """BASELINE configs[4] names a 60x file: the 24-Mb genome at twice the depth (9.6 M records) as one process and as two ranks
    sharing the GPU — byte-identical outputs, and the signal tables equal the CPU restatement's on the same file"""
    import socket
    import torch.multiprocessing as mp
    from tiddit_amd import __main__ as cli, synth_bam
    fx = load_fixture(golden_dir, "sv_e2e.json")
    P = dict(fx["params"], depth=60)
    contigs = synth_bam.wgs_contigs(P["total_mb"])
    fa, bam = str(tmp_path / "ref.fa"), str(tmp_path / "WGS.bam")
    seqs = synth_bam.write_fasta(fa, contigs, seed=P["fasta_seed"])
    info = synth_bam.write_wgs_sv_bam(bam, contigs, depth=60, read_len=P["read_len"], insert=P["insert"], insert_sd=P["insert_sd"], seed=P["seed"],
                                      sv_per_mb=P["sv_per_mb"], threads=min(16, os.cpu_count() or 1), ref_seqs=seqs)
    assert info["n_records"] > 1.9 * fx["n_records"]
    one = str(tmp_path / "one")
    argv = ["--sv", "--bam", bam, "--ref", fa, "--skip_assembly", "-s", str(P["n_reads_stats"])]
    cli.main(argv + ["-o", one])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = str(tmp_path / "two")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sv_rank, args=(r, 2, port, q, argv + ["-o", two], 40)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(120)
    assert all(isinstance(v, list) for v in res.values()), res
    for rel in ["_tiddit/discordants_WGS.tab", "_tiddit/splits_WGS.tab", "_tiddit/clips_WGS.fa", ".ploidies.tab", ".candidates.tab"] + \
               ["_tiddit/clips/%s.fa" % n for n, ln in contigs if ln >= P["min_contig"]]:
        assert open(two + rel, "rb").read() == open(one + rel, "rb").read(), rel
    from tiddit_amd import tiddit_stats
    lib = tiddit_stats.statistics(bam, fa, P["min_q"], 100000, P["n_reads_stats"])
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], lib["percentile_insert_size"], "WGS", P["min_contig"],
                                                                      P["min_anchor_len"], P["min_clip_len"], want_clips=False)
    assert open(one + "_tiddit/discordants_WGS.tab").read() == disc and open(one + "_tiddit/splits_WGS.tab").read() == split
    assert disc.count("\n") > 1.5 * fx["discordants_rows"]
    assert sum(1 for l in open(one + ".candidates.tab")) > 50


def test_library_statistics_on_the_device_equal_the_host_loop(run):
    """tiddit_stats.statistics with the sampling loop, the cut-off and numpy's mean / std / 99.9th percentile on the device
    (csrc/tdt_stats.hip) == the read-by-read C loop + numpy on the host, for cut-offs inside a batch, at a batch edge, of one read and
    beyond the file, with one batch and with many (the counters carry over); the fixture's cut-off is also the reference's own dictionary"""
    from tiddit_amd import bamio, tiddit_stats
    fx, bam, fa, contigs, out = run
    if fx["params"]["total_mb"] > 30:
        pytest.skip("the smaller files cover it")
    P = fx["params"]
    for chunk in (None, 3 << 20):
        for n in (1, 2, 1000, 77777, P["n_reads_stats"], 10**9):
            res = []
            for host in ("1", "0"):
                os.environ["TIDDIT_STATS_HOST"] = host
                if chunk:
                    os.environ["TIDDIT_INGEST_CHUNK"] = str(chunk)
                try:
                    res.append(tiddit_stats.statistics(bam, fa, P["min_q"], 100000, n))
                finally:
                    os.environ.pop("TIDDIT_STATS_HOST", None)
                    os.environ.pop("TIDDIT_INGEST_CHUNK", None)
                    bamio.set_carry(None)
            assert res[0] == res[1], (chunk, n, res)
            assert all(type(res[0][k]) is type(res[1][k]) for k in res[0]), (chunk, n)
            if n == P["n_reads_stats"]:
                for k, v in fx["library"].items():
                    assert res[1][k] == v, (k, res[1][k], v)


def test_switches_off_leave_the_same_files(run, tmp_path, monkeypatch):
    """the one-process job places the signal files from a writer thread beside the ploidy table and the clustering (the default the `run`
    fixture used); switched off — files written before tiddit_signal.main returns — every output is the same bytes"""
    from tiddit_amd import __main__ as cli
    fx, bam, fa, contigs, out = run
    if fx["params"]["total_mb"] > 30:
        pytest.skip("the 3-Mb and 24-Mb files are enough for this comparison")
    monkeypatch.setenv("TIDDIT_BACKGROUND_WRITES", "0")
    out2 = str(tmp_path / "seq")
    cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out2, "--skip_assembly", "-s", str(fx["params"]["n_reads_stats"])])
    for r in ["_tiddit/discordants_WGS.tab", "_tiddit/splits_WGS.tab", "_tiddit/clips_WGS.fa", ".ploidies.tab", ".candidates.tab"]:
        a, b = open(out + r, "rb").read(), open(out2 + r, "rb").read()
        assert a == b and a, r


def test_cov_cli_on_a_header_of_thousands_of_contigs(run, tmp_path, monkeypatch):
    """`tiddit --cov` on the GRCh38-shaped file: one row per bin of ALL 3 366 contigs (tiddit_coverage.pyx:10-21 creates an array per @SQ
    line, print_coverage walks them in header order) — the bins come back in one piece (tdt_cov_finish_all); the device-ingest and the
    host-ingest runs write the same bytes, and every contig's rows equal the oracle's update_coverage on the decoded reads"""
    import oracle
    from tiddit_amd import __main__ as cli, bamio
    fx, bam, fa, contigs, out = run
    if fx["params"].get("contig_table") != "grch38":
        pytest.skip("the other fixtures have 27 contigs")
    dev_out, host_out = str(tmp_path / "dev"), str(tmp_path / "host")
    cli.main(["--cov", "--bam", bam, "-o", dev_out, "-z", "500"])
    monkeypatch.setenv("TIDDIT_HOST_INGEST", "1")
    cli.main(["--cov", "--bam", bam, "-o", host_out, "-z", "500"])
    monkeypatch.delenv("TIDDIT_HOST_INGEST")
    text = open(dev_out + ".bed", "rb").read()
    assert text == open(host_out + ".bed", "rb").read()
    rows = text.decode().splitlines()
    assert len(rows) == 1 + sum(-(-l // 500) for _, l in contigs) and rows[0].startswith("#")
    # the oracle on the decoded reads, contig by contig (placed reads that pass the --cov filter, __main__.py:231-240)
    r = bamio.BamReader(bam)
    cols = {k: [] for k in ("tid", "pos", "end", "mapq", "flag")}
    for b in r.batches():
        for k in cols:
            cols[k].append(getattr(b, k))
    r.close()
    tid, pos, end, mapq, flag = (np.concatenate(cols[k]) for k in ("tid", "pos", "end", "mapq", "flag"))
    order = np.argsort(tid, kind="stable")
    cuts = np.searchsorted(tid[order], np.arange(len(contigs) + 1))
    at = 1
    for t, (name, ln) in enumerate(contigs):
        nb = -(-ln // 500)
        idx = order[cuts[t]:cuts[t + 1]]
        want, _ = oracle.coverage_stream(pos[idx], end[idx], mapq[idx], flag[idx], ln, 500, 20)      # the --cov filter (a3) + update_coverage
        got = np.array([float(x.split("\t")[3]) for x in rows[at:at + nb]])
        assert rows[at].split("\t")[0] == name and np.array_equal(got, want), name
        at += nb
