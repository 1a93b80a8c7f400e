"""world_size-2 gloo tests of the multi-GPU sharding path (CPU; the per-rank compute is the oracle,
standing in for the HIP kernels that the same code drives on the GPU box)."""
import os
import socket
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_buckets_lpt():
    from tiddit_amd.dist import shard_buckets
    sizes = [100, 1, 50, 50, 7, 0, 30]
    sh = shard_buckets(sizes, 3)
    assert sorted(sum(sh, [])) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in s) for s in sh]
    assert loads == [100, 80, 58]
    assert shard_buckets(sizes, 1) == [list(range(len(sizes)))]
    assert shard_buckets([], 4) == [[], [], [], []]


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    import oracle
    from tiddit_amd.dist import allgatherv, cluster_buckets_distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parts = allgatherv(torch.arange(rank * 3 + 1, dtype=torch.float64) + 10 * rank)
        assert [p.numel() for p in parts] == [r * 3 + 1 for r in range(world)]
        assert all(torch.equal(parts[r], torch.arange(r * 3 + 1, dtype=torch.float64) + 10 * r) for r in range(world))
        empty = allgatherv(torch.zeros(0, dtype=torch.float64))
        assert all(p.numel() == 0 for p in empty)
        rng = np.random.default_rng(5)           # same buckets on every rank
        sizes = [0, 40, 700, 3, 1200, 90, 1, 350]
        buckets = []
        for s in sizes:
            x = np.sort(rng.integers(0, max(10, s * 30), s))
            y = x + rng.integers(0, 900, s)
            buckets.append(np.stack([x, y], 1).astype(np.int64).reshape(s, 2))

        def cluster_local(ids):
            out = [oracle.dbscan_main(buckets[b], 300, 3) if sizes[b] else np.zeros(0) for b in ids]
            return torch.from_numpy(np.concatenate(out) if out else np.zeros(0))

        labels = cluster_buckets_distributed(sizes, cluster_local)
        for b, s in enumerate(sizes):
            want = oracle.dbscan_main(buckets[b], 300, 3) if s else np.zeros(0)
            assert np.array_equal(labels[b].numpy(), want), b
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_allgatherv_and_bucket_sharding_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _shard_truth(path, world):
    """Ground truth of a sharded read from the host decoder: per shard the records (by stream offset) that START in the
    shard's blocks, plus the seam offsets first_off / next_off that DeviceBamReader(shard=...) reports on the GPU."""
    import struct
    from tiddit_amd import bamio
    raw = open(path, "rb").read()
    starts, infl = [], []                       # per BGZF block: file offset, inflated offset
    o = u = 0
    while o < len(raw):
        starts.append(o)
        infl.append(u)
        bs = struct.unpack_from("<H", raw, o + 16)[0] + 1
        u += struct.unpack_from("<I", raw, o + bs - 4)[0]
        o += bs
    r = bamio.BamReader(path, batch_bytes=1 << 20)
    hdr = r.header_bytes
    rec, off, b0 = [], hdr, 0
    cols = {"pos": [], "end": [], "mapq": [], "flag": [], "tid": []}
    for b in r.batches():
        sizes = np.diff(np.concatenate([b.rec_off, [b.rec_off[-1] + 4 + int(np.frombuffer(bytes(b.raw[int(b.rec_off[-1]):int(b.rec_off[-1]) + 4]), "<u4")[0])]]))
        rec.append(off + np.concatenate([[0], np.cumsum(sizes[:-1])]))
        off += int(sizes.sum())
        for k in cols:
            cols[k].append(getattr(b, k))
    r.close()
    rec = np.concatenate(rec).astype(np.int64)
    cols = {k: np.concatenate(v) for k, v in cols.items()}
    fsize = len(raw)
    shards = []
    for rk in range(world):
        lo = bamio.find_block_start(path, fsize * rk // world)
        hi = fsize if rk == world - 1 else bamio.find_block_start(path, fsize * (rk + 1) // world)
        assert lo in starts or lo == fsize
        u_lo = infl[starts.index(lo)] if lo < fsize else u
        u_hi = infl[starts.index(hi)] if hi < fsize else u
        sel = (rec >= u_lo) & (rec < u_hi)
        first = int(rec[sel][0] - u_lo) if sel.any() else None
        nxt = rec[rec >= u_hi]
        shards.append(dict(sel=sel, empty=lo >= hi, first_off=first, next_off=int(nxt[0] - u_hi) if len(nxt) else 0))
    return cols, shards


def _cov_worker(rank, world, port, q, path):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    import oracle
    from tiddit_amd.dist import allreduce_bins, check_seams
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols, shards = _shard_truth(path, world)
        me = shards[rank]
        table = check_seams(me["first_off"], me["next_off"], me["empty"])
        assert len(table) == world
        try:                                                   # a shard that starts two bytes early must be caught
            check_seams(None if me["first_off"] is None else me["first_off"] + (2 if rank == 1 else 0), me["next_off"], me["empty"])
            caught = False
        except ValueError:
            caught = True
        assert caught
        # each rank bins its own records for every contig; the float64 all-reduce is exact
        LN = {0: 60000, 1: 40000, 2: 3000, 3: 500}
        parts, want = [], []
        for t, ln in LN.items():
            m = me["sel"] & (cols["tid"] == t)
            parts.append(oracle.coverage_stream(cols["pos"][m], cols["end"][m], cols["mapq"][m], cols["flag"][m], ln, 50, 20)[0])
            a = cols["tid"] == t
            want.append(oracle.coverage_stream(cols["pos"][a], cols["end"][a], cols["mapq"][a], cols["flag"][a], ln, 50, 20)[0])
        got = allreduce_bins(torch.from_numpy(np.concatenate(parts))).numpy()
        assert np.array_equal(got, np.concatenate(want)) and got.sum() > 0
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_bam_seams_and_exact_allreduce_gloo_world2(tmp_path):
    """one BAM, two ranks: byte-range shards of the BGZF blocks, seam agreement, and bins whose SUM all-reduce is bit-exact"""
    import torch.multiprocessing as mp
    from tiddit_amd import build, synth_bam
    build.build()
    path = str(tmp_path / "syn.bam")
    synth_bam.write_synthetic_bam(path, [("chr1", 60000), ("chr2", 40000), ("chrM", 3000), ("tiny", 500)], depth=6, seed=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cov_worker, args=(r, 2, port, q, path)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _bench_worker(rank, world, port, q):
    """the N-rank clustering step of bench.py (bench.shared_step over bench.shared_bucket_sizes), the oracle standing in for the
    device call of every rank's share"""
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    import bench
    import oracle
    from tiddit_amd import synth
    from tiddit_amd.dist import shard_buckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = bench.shared_bucket_sizes(30000)
        assert len(sizes) == 300 and 0.9 * 30000 < sizes.sum() <= 30000
        owned = shard_buckets(sizes, world)
        loads = [int(sum(sizes[b] for b in o)) for o in owned]
        assert max(loads) - min(loads) <= sizes.max()                 # bin packing by signal count
        pts = lambda b: synth.gen_points(int(sizes[b]), L=3_000_000, seed=1000 + b)
        calls = []

        def cluster_local(ids):
            calls.append(list(ids))
            out = [oracle.dbscan_main(pts(b), 500, 3) for b in ids if sizes[b]]
            return torch.from_numpy(np.concatenate(out) if out else np.zeros(0))

        from tiddit_amd.dist import split_gathered
        labels = split_gathered(sizes, *bench.shared_step(sizes, cluster_local, True))
        assert calls == [owned[rank]]                                 # a rank only ever clusters its own share ...
        for b in range(len(sizes)):                                   # ... and ends up with every bucket's labels
            want = oracle.dbscan_main(pts(b), 500, 3) if sizes[b] else np.zeros(0)
            assert np.array_equal(labels[b].numpy(), want), b
        one = split_gathered(sizes, *bench.shared_step(sizes, lambda ids: torch.from_numpy(np.concatenate([oracle.dbscan_main(pts(b), 500, 3) for b in ids if sizes[b]])), False))
        assert all(np.array_equal(one[b].numpy(), labels[b].numpy()) for b in range(len(sizes)))
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bench_shared_clustering_step_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---------------------------------------------------------------------------------------------------------------------
# one oversized bucket cut at posA gaps >= eps: planner, halo point and id re-basing against the oracle on
# the UNCUT bucket

def _oracle_cluster_buckets(buckets, epsilon, m, ctx=None, counts=False):
    """tiddit_cluster.cluster_buckets with the C oracle standing in for tdt_sort_dbscan_ex (same contract)"""
    import oracle
    labs, runs, last = [], [], []
    for b in buckets:
        b = np.asarray(b, dtype=np.int64).reshape(-1, 2) if len(b) else np.zeros((0, 2), np.int64)
        o = np.argsort(b[:, 0], kind="stable")
        d = np.ascontiguousarray(b[o])
        lab = np.empty(len(b))
        if len(b):
            xl, xid = oracle.x_coordinate_clustering(d, epsilon, m)
            yl, yid = oracle.y_coordinate_clustering(d, epsilon, m, xid, xl.copy())
            lab[o] = yl
        else:
            xid = yid = -1
        labs.append(lab)
        runs.append(xid + 1)
        last.append(yid)
    return (labs, np.array(runs, np.int64), np.array(last, np.int64)) if counts else labs


def _clumpy_bucket(rng, n, span, eps):
    """signals in clumps (clusters with several y sub-runs) over sparse noise: gaps >= eps exist between clumps"""
    centres = rng.integers(0, span, max(2, n // 40))
    x = np.concatenate([rng.choice(centres, n - n // 5) + rng.integers(0, eps * 2, n - n // 5), rng.integers(0, span, n // 5)])
    y = x + rng.choice([300, 5000, 90000], len(x)) + rng.integers(0, eps, len(x))
    p = rng.permutation(len(x))
    return np.stack([x[p], y[p]], 1).astype(np.int64)


def test_plan_bucket_cuts_are_legal_gaps():
    from tiddit_amd.dist import plan_bucket_cuts
    rng = np.random.default_rng(3)
    for eps in (1, 7.5, 120, 500):
        b = _clumpy_bucket(rng, 4000, 3_000_000, int(eps) + 1)
        ts, w = plan_bucket_cuts(b[:, 0], eps, 6)
        assert ts == sorted(set(ts)) and len(ts) >= 2 and w >= eps
        for t in ts:
            assert not ((b[:, 0] >= t) & (b[:, 0] < t + w)).any()           # nothing inside the gap ...
            assert (b[:, 0] < t).any() and (b[:, 0] >= t + w).any()         # ... and signals on both sides
    assert plan_bucket_cuts(np.arange(100), 5, 4) == ([], 0)                # no gap anywhere: not cut
    assert plan_bucket_cuts(np.array([1, 2, 3]), 1, 4) == ([], 0)
    assert plan_bucket_cuts(np.arange(0, 100000, 1000), 0, 4) == ([], 0)   # eps 0: nothing clusters, nothing to plan


def test_cut_pieces_rebased_equal_uncut_bucket(monkeypatch):
    """pieces (with their halo point) clustered independently + rebase_pieces == DBSCAN.main on the whole bucket, incl. the
    short-last-window quirk (DBSCAN.py:41-43) that a cut without the halo gets wrong"""
    import oracle
    from tiddit_amd import tiddit_cluster as tc
    monkeypatch.setattr(tc, "cluster_buckets", _oracle_cluster_buckets)
    rng = np.random.default_rng(11)
    halo_mattered = 0
    for trial in range(40):
        eps, m = int(rng.choice([40, 150, 500])), int(rng.choice([2, 3, 4, 6]))
        buckets = [_clumpy_bucket(rng, int(rng.integers(300, 2500)), 400_000, eps), _clumpy_bucket(rng, 60, 400_000, eps)]
        pieces = tc.plan_pieces(buckets, eps, 4, balance=0.5, min_cut=100)
        assert sum(1 for p in pieces if p[0] == 0) >= 3 and [p for p in pieces if p[0] == 1][0][2] is None
        labs, runs, last = tc.cluster_pieces_local(buckets, pieces, list(range(len(pieces))), eps, m)
        got = tc.assemble_pieces(buckets, pieces, labs, runs, last)
        want = _oracle_cluster_buckets(buckets, eps, m)
        for b in range(2):
            assert np.array_equal(got[b], want[b]), (trial, b, eps, m)
        # the same cut WITHOUT the halo point: pieces end one window early -> at least sometimes different labels
        bare = [(b, k, mem, None) for b, k, mem, _ in pieces]
        l2, r2, s2 = tc.cluster_pieces_local(buckets, bare, list(range(len(bare))), eps, m)
        halo_mattered += not np.array_equal(tc.assemble_pieces(buckets, bare, l2, r2, s2)[0], want[0])
    assert halo_mattered > 0


def _cut_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from tiddit_amd import tiddit_cluster as tc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def spy(buckets, epsilon, m, ctx=None, counts=False):
            calls.append(sum(len(b) for b in buckets))
            return _oracle_cluster_buckets(buckets, epsilon, m, ctx, counts)
        tc.cluster_buckets = spy
        rng = np.random.default_rng(21)                      # the same two-bucket job on every rank
        buckets = [_clumpy_bucket(rng, 6000, 2_000_000, 300), _clumpy_bucket(rng, 5000, 2_000_000, 300), np.zeros((0, 2), np.int64)]
        got = tc.cluster_buckets_sharded(buckets, 300, 3, min_cut=1000)
        want = _oracle_cluster_buckets(buckets, 300, 3)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        total = sum(len(b) for b in buckets)
        assert len(calls) == 1 and abs(calls[0] - total / world) < 0.15 * total      # two big buckets, balanced over the ranks by the cut
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_oversized_bucket_cut_gloo_world2_and_3():
    import torch.multiprocessing as mp
    for world in (2, 3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_cut_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(60)
        assert sorted(res) == [(r, "ok") for r in range(world)], res


# ---- `tiddit --sv` on N ranks behind the scan (tiddit_signal.share_and_write + tiddit_cluster.main_sharded): rows to the owner of
# their chrA in ONE all-to-all, every owner formats / places its blocks and clusters / regroups its buckets; rank 0 only puts the
# finished candidates together.  The two device steps are the host stand-ins of tests/sigtab_common.py.

def _sv_tables_worker(rank, world, port, q, bam, fixture, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import json
    import torch.distributed as dist
    from oracle import cluster_oracle
    from sigtab_common import fill_tables, oracle_labels
    from tiddit_amd import bamio, tiddit_cluster, tiddit_signal
    from tiddit_amd.sigtab import SignalTables
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = json.load(open(fixture))
        P = fx["params"]
        rd = bamio.BamReader(bam, batch_bytes=4 << 20)
        batches = list(rd.batches())
        rd.close()
        mine = batches[len(batches) * rank // world:len(batches) * (rank + 1) // world]        # this rank's share of the file, in file order
        big = [ln >= P["min_contig"] for ln in rd.lengths]
        scanned = SignalTables(rd.references, rd.lengths, P["min_contig"])
        max_ins = fx["library"]["percentile_insert_size"]
        fill_tables(scanned, mine, rd.references, big, P["min_q"], max_ins, P["min_anchor_len"], P["min_clip_len"])
        prefix = os.path.join(outdir, "n%d" % world)
        chromosomes = [n for n, ok in zip(rd.references, big) if ok]
        merged, owner = tiddit_signal.share_and_write(scanned, chromosomes, prefix, "WGS")
        assert tiddit_signal.written_tables(prefix + "_tiddit/discordants_WGS.tab", prefix + "_tiddit/splits_WGS.tab") is merged
        st = merged.stats()

        class Pageable:
            def take(self, name, n, dtype):
                return np.zeros(n, dtype=dtype)
        tiddit_cluster._POOL = Pageable()
        tiddit_cluster.cluster_columns_device = oracle_labels
        cand = tiddit_cluster.main_sharded(prefix, rd.references, dict(zip(rd.references, rd.lengths)), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                                           max_ins, P["min_contig"], True, P["min_reads"])
        res = {"rows": st["discordant_rows"] + st["split_rows"], "stages": sorted(tiddit_cluster.STAGE_SECONDS)}
        if rank == 0:
            res["canonical"] = cluster_oracle.canonical(cand)
            res["summary"] = cluster_oracle.summary(cand)
        else:
            assert cand is None
        # ONE rank loses its tables (as if its files had been touched): the native-vs-text choice is agreed on by all ranks (the two
        # branches run different collectives), so EVERY rank parses the text now, and rank 0 still returns the same candidates
        dist.barrier()
        if rank == world - 1:
            tiddit_signal._forget_tables()

        def all_buckets_here(buckets, epsilon, m, **kw):
            import oracle
            out = []
            for b in buckets:
                order = np.argsort(b[:, 0], kind="stable")
                lab = np.empty(len(b))
                lab[order] = oracle.dbscan_main(b[order], epsilon, m)
                out.append(lab)
            return out
        tiddit_cluster.cluster_buckets_sharded = all_buckets_here
        again = tiddit_cluster.main_sharded(prefix, rd.references, dict(zip(rd.references, rd.lengths)), ["WGS"], fx["library"]["mp"], fx["epsilon"], P["m"],
                                            max_ins, P["min_contig"], True, P["min_reads"])
        res["text_stages"] = sorted(tiddit_cluster.STAGE_SECONDS)
        if rank == 0:
            res["text_canonical"] = cluster_oracle.canonical(again)
        else:
            assert again is None
        q.put((rank, res))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sv_signal_tables_on_n_ranks_gloo_world2_and_3(tmp_path):
    """the fixture's 3-Mb WGS file dealt to 2 / 3 ranks: discordants / splits .tab, every clip FASTA and the candidates dictionary are
    the single-process run's (= the fixture's checksums); every owner holds rows, none of them all"""
    import hashlib
    import json
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from sv_e2e_common import load_fixture, materialise
    golden = os.path.join(REPO, "tests", "golden")
    fx = load_fixture(golden, "sv_e2e_small.json")
    bam, fa, contigs = materialise(fx, str(tmp_path), threads=4)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    for world in (2, 3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_sv_tables_worker, args=(r, world, port, q, bam, os.path.join(golden, "sv_e2e_small.json"), str(tmp_path))) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(60)
        assert all(isinstance(v, dict) for v in res.values()), res
        prefix = str(tmp_path / ("n%d" % world))
        assert sha(prefix + "_tiddit/discordants_WGS.tab") == fx["discordants_sha256"]
        assert sha(prefix + "_tiddit/splits_WGS.tab") == fx["splits_sha256"]
        assert sha(prefix + "_tiddit/clips_WGS.fa") == fx["clips_sha256"]
        each = b"".join(open(prefix + "_tiddit/clips/%s.fa" % n, "rb").read() for n, ln in contigs if ln >= fx["params"]["min_contig"])
        assert hashlib.sha256(each).hexdigest() == fx["clips_sha256"]
        assert res[0]["summary"] == json.loads(json.dumps(fx["candidates"])) and hashlib.sha256(res[0]["canonical"].encode()).hexdigest() == fx["candidates_sha256"]
        rows = [res[r]["rows"] for r in range(world)]
        assert all(x > 0 for x in rows) and max(rows) < sum(rows)
        assert all("candidates to rank 0" in res[r]["stages"] and "parse .tab" not in res[r]["stages"] for r in range(world))
        assert all("parse .tab" in res[r]["text_stages"] and "candidates to rank 0" not in res[r]["text_stages"] for r in range(world))
        assert res[0]["text_canonical"] == res[0]["canonical"]


def _a2a_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from tiddit_amd import dist as tdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank s has (s + 1) * 1000 * d bytes for rank d (nothing for rank 0, nothing for itself when s is odd): every byte says who sent it to whom
        def part(s, d):
            n = 0 if d == 0 or (s == d and s % 2) else (s + 1) * 1000 * d
            return np.full(n, 16 * s + d, dtype=np.uint8)
        for cap in (None, 777, 1):                               # one message per piece; pieces cut into several (TIDDIT_WIRE_MAX_BYTES)
            if cap is None:
                os.environ.pop("TIDDIT_WIRE_MAX_BYTES", None)
            else:
                os.environ["TIDDIT_WIRE_MAX_BYTES"] = str(cap)
            scale = 1 if cap != 1 else 100                         # (byte-sized messages: smaller payloads)
            got = tdist.alltoall_bytes([part(rank, d)[::scale] for d in range(world)])
            assert len(got) == world
            for s in range(world):
                assert np.array_equal(np.asarray(got[s]), part(s, rank)[::scale]), (rank, s, cap)
            # ragged byte strings to rank 1 (rank 2 sends nothing), and a pickled object from rank 2 to everybody
            blobs = tdist.gather_bytes(bytes([65 + rank]) * (0 if rank == 2 else 1500 * (rank + 1) // scale), dst=1)
            assert (blobs is None) == (rank != 1)
            if rank == 1:
                assert blobs == [bytes([65 + r]) * (0 if r == 2 else 1500 * (r + 1) // scale) for r in range(world)]
            obj = tdist.broadcast_object({"rank": rank, "payload": list(range(400 // scale)), "none": None} if rank == 2 else None, src=2)
            assert obj == {"rank": 2, "payload": list(range(400 // scale)), "none": None}
        os.environ.pop("TIDDIT_WIRE_MAX_BYTES", None)
        sizes = tdist.allgather_i64([rank, 10 * rank, -rank])
        assert sizes.shape == (world, 3) and sizes[:, 1].tolist() == [10 * r for r in range(world)]
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_alltoall_bytes_and_contig_owners_gloo_world3():
    """the one exchange of the N-rank signal tables: ragged and empty payloads arrive in rank order; the owner map is a pure function"""
    import torch.multiprocessing as mp
    from tiddit_amd.dist import contig_owners
    lengths = [248, 242, 198, 0.5, 190, 182, 171, 16e-3, 159]
    kept = [ln >= 1 for ln in lengths]
    for world in (1, 2, 3, 8):
        own = contig_owners(lengths, kept, world)
        assert own.dtype == np.int32 and len(own) == len(lengths) and set(own.tolist()) <= set(range(world))
        load = [sum(l for l, o, k in zip(lengths, own.tolist(), kept) if k and o == r) for r in range(world)]
        assert max(load) - min(l for l in load if l or world <= 7) <= max(lengths) or world >= 7      # LPT: no rank more than one contig ahead
        assert np.array_equal(own, contig_owners(lengths, kept, world))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_a2a_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(v == "ok" for v in res.values()), res
