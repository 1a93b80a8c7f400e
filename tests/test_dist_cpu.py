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
