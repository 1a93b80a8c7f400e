"""CPU side of BASELINE configs[3]: the oracle pieces of the --sv path are pinned to the fixture the compiled reference produced
(tests/golden/sv_e2e_small.json), and the product's host stages that need no GPU are checked against the same numbers."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from oracle import cluster_oracle, signal_oracle

from sv_e2e_common import load_fixture, materialise


def h(t):
    return hashlib.sha256(t.encode()).hexdigest()


def _jsonable(c):               # the encoding tests/golden/make_golden.py stores candidates dictionaries in
    if isinstance(c, dict):
        return {str(k): _jsonable(v) for k, v in c.items()}
    if isinstance(c, set):
        return {"__set__": sorted(_jsonable(x) for x in c)}
    if isinstance(c, (list, tuple)):
        return [_jsonable(x) for x in c]
    if isinstance(c, (np.integer,)):
        return int(c)
    if isinstance(c, (np.floating,)):
        return float(c)
    return c


@pytest.fixture(scope="module")
def small(golden_dir, tmp_path_factory):
    fx = load_fixture(golden_dir, "sv_e2e_small.json")
    d = str(tmp_path_factory.mktemp("sv_small"))
    bam, fa, contigs = materialise(fx, d, threads=4)
    return fx, bam, fa, contigs, d


def test_cluster_oracle_reproduces_the_reference_cases(golden_dir, tmp_path):
    """oracle/cluster_oracle.py against tiddit_cluster.main's own outputs (six hand-built .tab pairs, tests/golden/cluster.json)"""
    g = json.load(open(os.path.join(golden_dir, "cluster.json")))
    for t in g["find_discordant_pos"]:
        frag = ["q", "1", "1", "100", "250", t["revA"], "900", "1050", t["revB"]]
        assert list(cluster_oracle.find_discordant_pos(frag, t["is_mp"])) == t["out"]
    for k, case in enumerate(g["cases"]):
        prefix = str(tmp_path / ("c%d" % k))
        os.makedirs(prefix + "_tiddit")
        open(prefix + "_tiddit/discordants_S.tab", "w").write("".join(l + "\n" for l in case["discordants_tab"]))
        open(prefix + "_tiddit/splits_S.tab", "w").write("".join(l + "\n" for l in case["splits_tab"]))
        cand = cluster_oracle.main(prefix, case["chromosomes"], case["contig_length"], ["S"], case["is_mp"], case["epsilon"], case["m"],
                                   case["max_ins_len"], case["min_contig"], True, case["min_reads"])
        # same keys, same values AND same insertion order (the order defines the VCF SV ids downstream)
        assert json.dumps(_jsonable(cand)) == json.dumps(case["candidates"]), k


def test_signal_oracle_file_scale_equals_per_read_literal(small):
    """signal_main_file (C record walk + C predicate chain, only marked reads parsed) == signal_main (every read a Python object)"""
    fx, bam, fa, contigs, d = small
    hdr, reads = signal_oracle.parse_bam(bam)
    max_ins = fx["library"]["percentile_insert_size"]
    P = fx["params"]
    wcov, wdisc, wsplit, wclips, weach = signal_oracle.signal_main(hdr, reads, P["min_q"], int(max_ins), "WGS", P["min_contig"], P["min_anchor_len"], P["min_clip_len"])
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], max_ins, "WGS", P["min_contig"], P["min_anchor_len"], P["min_clip_len"])
    assert n == len(reads) == fx["n_records"]
    assert disc == wdisc and split == wsplit and clips == wclips and each == weach
    assert list(cov) == list(wcov) and all(np.array_equal(cov[c], wcov[c]) for c in cov)
    # ... and both equal what the fixture was computed from
    assert h(disc) == fx["discordants_sha256"] and h(split) == fx["splits_sha256"] and h(clips) == fx["clips_sha256"]
    for c in cov:
        assert hashlib.sha256(cov[c].astype("<f8").tobytes()).hexdigest() == fx["coverage_sha256"][c], c
    assert fx["discordants_rows"] > 300 and fx["splits_rows"] > 100 and fx["clips_entries"] > 500


def test_cluster_oracle_reproduces_the_reference_candidates(small, tmp_path):
    """the restatement on the e2e .tab files == the compiled tiddit_cluster.main (whole nested dictionary, insertion order included)"""
    fx, bam, fa, contigs, d = small
    P = fx["params"]
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], fx["library"]["percentile_insert_size"], "WGS",
                                                                      P["min_contig"], P["min_anchor_len"], P["min_clip_len"], want_clips=False)
    prefix = str(tmp_path / "o")
    os.makedirs(prefix + "_tiddit")
    open(prefix + "_tiddit/discordants_WGS.tab", "w").write(disc)
    open(prefix + "_tiddit/splits_WGS.tab", "w").write(split)
    names = [n_ for n_, _ in contigs]
    cand = cluster_oracle.main(prefix, names, dict(contigs), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                               fx["library"]["percentile_insert_size"], P["min_contig"], True, P["min_reads"])
    assert cluster_oracle.summary(cand) == fx["candidates"]
    assert h(cluster_oracle.canonical(cand)) == fx["candidates_sha256"]
    # the planted events come back: every DEL/DUP/INV/BND with enough support has a candidate within the clustering distance
    found = 0
    for ev in fx["events"]:
        for row in fx["candidates"]:
            if {row[0], row[1]} == {ev["chrA"], ev["chrB"]} and min(abs(row[3] - ev["posA"]) + abs(row[4] - ev["posB"]), abs(row[3] - ev["posB"]) + abs(row[4] - ev["posA"])) < 1000:
                found += 1
                break
    assert found >= 0.8 * len(fx["events"])


def test_product_statistics_equal_the_reference(small, monkeypatch):
    """tiddit_amd.tiddit_stats.statistics (vectorised on the host-decoded arrays) == tiddit_stats.py of the reference (fixture)"""
    fx, bam, fa, contigs, d = small
    monkeypatch.setenv("TIDDIT_HOST_INGEST", "1")
    from tiddit_amd import tiddit_stats
    for n_reads in (fx["params"]["n_reads_stats"],):
        lib = tiddit_stats.statistics(bam, fa, fx["params"]["min_q"], 100000, n_reads)
        for k, v in fx["library"].items():
            assert lib[k] == v, (k, lib[k], v)


def test_bounded_contig_sample_equals_the_full_restatement(small):
    """signal_oracle.signal_main_sample — a run of contigs located by a binary search over the BGZF blocks, no index, nothing else
    inflated (what bench.py times on a 3-Gb BAM instead of the whole file) — gives, for those contigs, exactly what the full
    restatement gives: coverage, clip FASTA, and the rows of both tables whose contigs all lie in the run; one process and several"""
    fx, bam, fa, contigs, d = small
    P = fx["params"]
    max_ins = fx["library"]["percentile_insert_size"]
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], max_ins, "WGS", P["min_contig"], P["min_anchor_len"], P["min_clip_len"])
    header, sq = signal_oracle.read_header(bam)
    assert [c["SN"] for c in sq] == [c for c, _ in contigs]
    offs = signal_oracle.block_offsets(bam)
    assert offs[0] == 0 and offs[-1] == os.path.getsize(bam) - 28 + 28       # hops end exactly at the end of the file
    big = [t for t, (c, ln) in enumerate(contigs) if ln >= P["min_contig"]]
    for tids, procs in ((big[-5:], 1), (big[-5:], 3), (big[3:9], 2), (big[:2], 1)):
        names = {contigs[t][0] for t in tids}
        scov, sdisc, ssplit, sclips, nrec = signal_oracle.signal_main_sample(bam, tids, P["min_q"], max_ins, P["min_anchor_len"], P["min_clip_len"], procs)
        for t in tids:
            c = contigs[t][0]
            assert np.array_equal(scov[c], cov[c]), c
            assert sclips[c] == each[c], c
        keep = lambda text: "".join(l + "\n" for l in text.splitlines() if l.split("\t")[1] in names and l.split("\t")[2] in names)
        # (a split row needs one read only: the sample also has rows whose second contig lies outside the run — not comparable, dropped)
        assert sdisc == keep(disc) and keep(ssplit) == keep(split), (tids, procs)
        assert nrec > 0
    assert len(keep(disc)) > 0


def test_statistics_prefix_restatement_is_pinned(small):
    """oracle/signal_oracle.statistics_prefix (the CPU leg's library statistics: inflates only the sampled prefix of the file) equals
    the `library` dictionary the reference's own tiddit_stats.py produced for this file, at the fixture's cut-off and at smaller ones"""
    fx, bam, fa, contigs, d = small
    P = fx["params"]
    lib = signal_oracle.statistics_prefix(bam, P["min_q"], 100000, P["n_reads_stats"])
    for k, v in fx["library"].items():
        assert lib[k] == v, (k, lib[k], v)
    from tiddit_amd import tiddit_stats                     # (host path of the product: the same numbers for other cut-offs)
    os.environ["TIDDIT_HOST_INGEST"] = "1"
    try:
        for n in (1, 1000, 77777):
            a = signal_oracle.statistics_prefix(bam, P["min_q"], 100000, n)
            b = tiddit_stats.statistics(bam, fa, P["min_q"], 100000, n)
            assert a == {k: b[k] for k in a}, n
    finally:
        del os.environ["TIDDIT_HOST_INGEST"]


# ---- a header shaped like the GRCh38 analysis set's (tests/golden/sv_e2e_grch38.json: 3 366 contigs)

@pytest.fixture(scope="module")
def grch38(golden_dir, tmp_path_factory):
    fx = load_fixture(golden_dir, "sv_e2e_grch38.json")
    d = str(tmp_path_factory.mktemp("sv_grch38"))
    bam, fa, contigs = materialise(fx, d, threads=4)
    return fx, bam, fa, contigs, d


def test_grch38_shaped_header_host_side(grch38):
    """The contig table the reference loops over on a human sample (tiddit_signal.pyx:246-259, tiddit_cluster.pyx:140-147,
    tiddit_coverage.pyx:10-21): 3 366 @SQ lines with alt / decoy / HLA names (`*`, `:`), a header larger than three BGZF blocks,
    3 108 contigs below --min_contig.  Host side: the header parse across blocks, the statistics of the reference's own
    tiddit_stats.py (fixture), the per-contig coverage dictionary of create_coverage, and the contig-to-rank maps of the N-rank job."""
    fx, bam, fa, contigs, d = grch38
    from tiddit_amd import bamio, dist as tdist, tiddit_coverage
    assert len(contigs) == 3366 and len({n for n, _ in contigs}) == 3366
    assert sum(1 for n, _ in contigs if "*" in n and ":" in n) > 400 and min(l for _, l in contigs) == 970
    r = bamio.BamReader(bam)
    assert r.references == [n for n, _ in contigs] and r.lengths == [l for _, l in contigs]
    assert r.header_bytes > 3 * 0xff00                           # the header alone spans four BGZF blocks
    n, tail, tids = 0, 0, set()
    for b in r.batches():
        n += len(b)
        tail += int((b.tid < 0).sum())
        tids.update(np.unique(b.tid).tolist())
    r.close()
    assert n == fx["n_records"] and tail > 0 and len(tids) > 1500          # reads on 1 800 contigs, and the unplaced tail
    P = fx["params"]
    big = [n_ for n_, l in contigs if l >= P["min_contig"]]
    assert list(fx["coverage_sha256"]) == big and 200 < len(big) < 300
    assert any("*" in n_ for n_ in big)                          # HLA-DRB1 alleles above --min_contig: their names go into file names and rows
    # create_coverage (tiddit_coverage.pyx:10-21): one array per contig, ceil(LN / bin) bins
    header = {"SQ": [{"SN": n_, "LN": l} for n_, l in contigs]}
    cov, lens = tiddit_coverage.create_coverage(header, 50)
    assert list(cov) == [n_ for n_, _ in contigs] and all(len(cov[n_]) == -(-l // 50) for n_, l in contigs)
    # the string order of the names (chrA < chrB, tiddit_signal.pyx:213) is not the header order
    names = [n_ for n_, _ in contigs]
    assert sorted(names) != names
    lens = [l for _, l in contigs]
    kept = [l >= P["min_contig"] for l in lens]
    for world in (2, 3, 8):
        owners = tdist.contig_owners(lens, kept, world)
        assert len(owners) == len(names) and set(owners.tolist()) == set(range(world))
        load = [sum(l for l, k, o in zip(lens, kept, owners) if k and o == r_) for r_ in range(world)]
        assert max(load) < 1.5 * sum(load) / world               # 258 kept contigs packed evenly; the 3 108 others own nothing
        parts = tdist.shard_contigs(lens, world)                 # the histogram path packs all 3 366
        assert sorted(i for p_ in parts for i in p_) == list(range(len(lens)))


def test_grch38_shaped_restatements_equal_the_reference(grch38, tmp_path):
    """the restatements on this file == the fixture made by the compiled reference: signal tables (sha), candidates of
    tiddit_cluster.main (whole dictionary), library statistics of tiddit_stats.py through the host decode"""
    fx, bam, fa, contigs, d = grch38
    P = fx["params"]
    cov, disc, split, clips, each, n = signal_oracle.signal_main_file(bam, P["min_q"], fx["library"]["percentile_insert_size"], "WGS",
                                                                      P["min_contig"], P["min_anchor_len"], P["min_clip_len"], want_clips=False)
    assert h(disc) == fx["discordants_sha256"] and h(split) == fx["splits_sha256"] and n == fx["n_records"]
    prefix = str(tmp_path / "o")
    os.makedirs(prefix + "_tiddit")
    open(prefix + "_tiddit/discordants_WGS.tab", "w").write(disc)
    open(prefix + "_tiddit/splits_WGS.tab", "w").write(split)
    names = [n_ for n_, _ in contigs]
    cand = cluster_oracle.main(prefix, names, dict(contigs), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                               fx["library"]["percentile_insert_size"], P["min_contig"], True, P["min_reads"])
    assert cluster_oracle.summary(cand) == fx["candidates"] and h(cluster_oracle.canonical(cand)) == fx["candidates_sha256"]
