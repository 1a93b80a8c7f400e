"""GPU parity tests: the HIP path (through the C ABI of libtiddit_hip.so) against the CPU oracle and
the golden vectors captured from the real reference.  Bit-exact everywhere (integer / exact-float
work).  Run on the MI355X box: python -m pytest tests -m gpu"""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from tiddit_amd import synth

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def nat():
    from tiddit_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def ctx(nat):
    return nat.default_context(0)


@pytest.fixture(scope="module")
def cov(ctx):
    from tiddit_amd import tiddit_coverage
    return tiddit_coverage


@pytest.fixture(scope="module")
def gcmod(ctx):
    from tiddit_amd import tiddit_gc
    return tiddit_gc


@pytest.fixture(scope="module")
def db(ctx):
    from tiddit_amd import DBSCAN
    return DBSCAN


# ------------------------------------------------------------------------------------ coverage
def test_native_library_is_loaded(nat):
    assert nat.load().tdt_version() >= 100
    maps = open("/proc/self/maps").read()
    assert "libtiddit_hip.so" in maps


def test_coverage_kat(cov, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "coverage.json")))
    for k in g["kat"]:
        arr, ebs = cov.create_coverage({"SQ": [{"SN": "c", "LN": k["LN"]}]}, k["bin"], "c")
        r = cov.update_coverage(k["s"], k["e"], k["bin"], arr, ebs)
        assert r is arr
        assert arr.tolist() == k["bins"], k
    for s in g["shapes"]:
        d, eb = cov.create_coverage({"SQ": [{"SN": "a", "LN": s["LN"]}, {"SN": "b", "LN": 77}]}, s["bin"])
        assert (len(d["a"]), eb["a"], len(d["b"]), eb["b"]) == (s["nbins"], s["end_bin_size"], s["nbins_b"], s["end_bin_size_b"])
        h = cov.CoverageHistogram([("a", s["LN"]), ("b", 77)], s["bin"])
        assert h.nbins("a") == (s["nbins"], s["end_bin_size"]) and h.nbins("b") == (s["nbins_b"], s["end_bin_size_b"])
        h.close()


def test_coverage_random_golden(cov, golden_dir):
    z = np.load(os.path.join(golden_dir, "coverage_random.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        LN, b = z["c%d_meta" % c].tolist()
        s, e = z["c%d_start" % c], z["c%d_end" % c]
        h = cov.CoverageHistogram([("c", LN)], b)
        h.push("c", s, e, np.full(len(s), 60, np.uint8), np.zeros(len(s), np.uint16), 0)
        got = h.finish("c")
        assert h.kept() == len(s)
        h.close()
        assert np.array_equal(got, z["c%d_bins" % c]), (c, LN, b)


@pytest.mark.parametrize("tag", ["cov", "sv"])
def test_coverage_config1_stream(cov, golden_dir, tag):
    g = json.load(open(os.path.join(golden_dir, "coverage.json")))["config1"][tag]
    start, end, mapq, flag = synth.gen_reads(1_000_000, 10)
    h = cov.CoverageHistogram([("chrS", 1_000_000)], g["bin"])
    h.push("chrS", start, end, mapq, flag, g["q"])
    bins = h.finish("chrS")
    assert h.kept() == g["kept"]
    assert sha(bins.astype("<f8")) == g["bins_sha256"]
    assert np.array_equal(bins, np.load(os.path.join(golden_dir, "config1_bins_%s.npy" % tag)))
    # pushing in two halves, out of order, gives the same bins (exact integer accumulation)
    h.reset()
    half = len(start) // 2 + 3
    h.push("chrS", start[half:], end[half:], mapq[half:], flag[half:], g["q"])
    h.push("chrS", start[:half], end[:half], mapq[:half], flag[:half], g["q"])
    assert np.array_equal(h.finish("chrS"), bins)
    h.close()


def test_coverage_print_matches_golden(cov, golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "coverage.json")))["config1"]["cov"]
    start, end, mapq, flag = synth.gen_reads(1_000_000, 10)
    hdr = {"SQ": [{"SN": "chrS", "LN": 1_000_000}]}
    d, eb = cov.create_coverage(hdr, 500)
    d["chrS"] = cov.update_coverage_batch(start, end, mapq, flag, 20, 500, d["chrS"], eb["chrS"])
    for ft in ("bed", "wig"):
        p = str(tmp_path / ("o." + ft))
        cov.print_coverage(d, hdr, 500, ft, p)
        assert hashlib.sha256(open(p, "rb").read()).hexdigest() == g[ft + "_sha256"]


@pytest.mark.parametrize("LN,z,depth,seed", [(3_000_000, 500, 30, 1), (3_000_000, 50, 30, 2), (1_000_003, 333, 60, 3),
                                              (5_000_000, 1000, 8, 4), (200_000, 1, 3, 5), (700_001, 4096, 40, 6)])
def test_coverage_stream_vs_oracle(cov, LN, z, depth, seed):
    start, end, mapq, flag = synth.gen_reads(LN, depth, seed=seed)
    want, kept = oracle.coverage_stream(start, end, mapq, flag, LN, z, 5)
    h = cov.CoverageHistogram([("c", LN)], z)
    h.push("c", start, end, mapq, flag, 5)
    got = h.finish("c")
    assert h.kept() == kept
    h.close()
    assert np.array_equal(got, want)


def test_coverage_long_and_unsorted_reads(cov):
    rng = np.random.default_rng(9)
    LN, z, n = 2_000_000, 50, 200_000
    start = rng.integers(0, LN - 1, n)                      # NOT sorted
    span = np.where(rng.random(n) < 0.05, rng.integers(1, 60_000, n), rng.integers(1, 400, n))  # ONT-like tails
    end = np.minimum(start + span, LN)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    flag = np.where(rng.random(n) < 0.1, 0x400, 0).astype(np.uint16)
    want, kept = oracle.coverage_stream(start, end, mapq, flag, LN, z, 20)
    h = cov.CoverageHistogram([("c", LN)], z)
    h.push("c", start, end, mapq, flag, 20)
    got = h.finish("c")
    assert h.kept() == kept
    assert np.array_equal(got, want)
    h.close()


def test_coverage_multi_contig_and_last_bin(cov):
    rng = np.random.default_rng(10)
    contigs = [("a", 1137), ("b", 100_000), ("c", 99_999), ("d", 1), ("e", 50_000)]
    h = cov.CoverageHistogram(contigs, 500)
    want = {}
    for name, LN in contigs:
        n = 4000
        start = np.sort(rng.integers(0, LN, n))
        end = np.minimum(start + rng.integers(1, 700, n), LN)      # many reads end exactly at LN
        mapq = np.full(n, 30, np.uint8)
        flag = np.zeros(n, np.uint16)
        want[name], _ = oracle.coverage_stream(start, end, mapq, flag, LN, 500, 20)
        h.push(name, start, end, mapq, flag, 20)
    for name, LN in contigs:
        assert np.array_equal(h.finish(name), want[name]), name
    h.close()


def test_coverage_out_of_range_raises(cov, nat):
    arr, ebs = cov.create_coverage({"SQ": [{"SN": "c", "LN": 1000}]}, 500, "c")
    with pytest.raises(IndexError):
        cov.update_coverage(900, 1200, 500, arr, ebs)
    h = cov.CoverageHistogram([("c", 1000)], 500)
    h.push("c", [10], [5], [60], [0], 0)   # end <= start
    with pytest.raises(nat.TdtError):
        h.finish("c")
    h.close()


def test_coverage_device_pointers_and_misalignment(cov, ctx):
    torch = pytest.importorskip("torch")
    LN, z = 4_000_000, 500
    start, end, mapq, flag = synth.gen_reads(LN, 20, seed=77)
    want, _ = oracle.coverage_stream(start[3:], end[3:], mapq[3:], flag[3:], LN, z, 20)
    dev = torch.device("cuda:0")
    ts = torch.from_numpy(start.astype(np.int32)).to(dev)
    te = torch.from_numpy(end.astype(np.int32)).to(dev)
    tm = torch.from_numpy(mapq).to(dev)
    tf = torch.from_numpy(flag.view(np.int16)).to(dev)
    out = torch.empty(len(want), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    h = cov.CoverageHistogram([("c", LN)], z)
    # offset by 3 elements: none of the arrays is vector-load aligned any more
    n = len(start) - 3
    h.push_device("c", ts.data_ptr() + 12, te.data_ptr() + 12, tm.data_ptr() + 3, tf.data_ptr() + 6, n, 20)
    h.finish_device("c", out.data_ptr())
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), want)
    h.reset()
    want0, _ = oracle.coverage_stream(start, end, mapq, flag, LN, z, 20)
    h.push_device("c", ts.data_ptr(), te.data_ptr(), tm.data_ptr(), tf.data_ptr(), len(start), 20)
    h.finish_device("c", out.data_ptr())
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), want0)
    h.close()


# ------------------------------------------------------------------------------------------ gc
def test_gc_kat(gcmod, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "gc.json")))
    for k in g["kat"]:
        out = gcmod.binned_gc_array(k["seq"].encode(), k["bin"], k["n_cutoff"])
        assert out.dtype == np.int8 and out.tolist() == k["out"], k


def test_gc_random_golden(gcmod, golden_dir):
    z = np.load(os.path.join(golden_dir, "gc_random.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        L, b, seed = z["c%d_meta" % c].tolist()
        seq = synth.gen_sequence(L, seed=seed, n_frac=0.08)
        out = gcmod.binned_gc_array(seq, b, float(z["c%d_cut" % c][0]))
        assert np.array_equal(out, z["c%d_out" % c]), (c, L, b)


@pytest.mark.parametrize("L,b,cut", [(10_000_019, 50, 0.5), (3_000_001, 500, 0.5), (2_000_000, 64, 0.1), (1_000_000, 2048, 0.5),
                                     (1_000_000, 2049, 0.5), (5_000_000, 100_000, 0.3), (999_983, 13, 0.0), (64, 50, 0.5),
                                     (1_000_000, 1_000_000, 0.5), (1_000_001, 1_000_000, 1.0)])
def test_gc_vs_oracle(gcmod, L, b, cut):
    seq = synth.gen_sequence(L, seed=L % 97, n_frac=0.1)
    assert np.array_equal(gcmod.binned_gc_array(seq, b, cut), oracle.binned_gc(seq, b, cut))


def test_gc_fasta_entry_points(gcmod, tmp_path):
    fa = tmp_path / "ref.fa"
    seqs = {"chrA": synth.gen_sequence(12_345, seed=1), "chrB": synth.gen_sequence(700, seed=2), "chrC": synth.gen_sequence(61, seed=3)}
    with open(fa, "w") as f:
        for name, s in seqs.items():
            f.write(">%s some description\n" % name)
            t = s.tobytes().decode()
            for i in range(0, len(t), 60):
                f.write(t[i:i + 60] + "\n")
    r = gcmod.binned_gc(str(fa), "chrB", 50, 0.5)
    assert r[0] == "chrB" and np.array_equal(r[1], oracle.binned_gc(seqs["chrB"], 50, 0.5))
    d = gcmod.main(str(fa), list(seqs), 4, 50, 0.5)
    for name, s in seqs.items():
        assert np.array_equal(d[name], oracle.binned_gc(s, 50, 0.5)), name


# -------------------------------------------------------------------------------------- dbscan
def test_dbscan_kat(db, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dbscan.json")))
    for k in g["kat"]:
        data = np.array(k["data"], dtype=np.int64)
        xl, xid = db.x_coordinate_clustering(data, k["eps"], k["m"])
        assert xl.tolist() == k["x"] and xid == k["x_id"], k
        yl, yid = db.y_coordinate_clustering(data, k["eps"], k["m"], xid, xl.copy())
        assert yl.tolist() == k["y"] and yid == k["y_id"], k
        assert db.main(data, k["eps"], k["m"]).tolist() == k["y"]
    # the reference's own commented toy (DBSCAN.py:132-133) given as a list of lists
    assert db.main([[1, 2], [1, 2], [1, 2], [10, 11]], 0.1, 2).tolist() == [0.0, 0.0, -1.0, -1.0]


def test_dbscan_random_golden(db, golden_dir):
    z = np.load(os.path.join(golden_dir, "dbscan_random.npz"))
    n = len([k for k in z.files if k.endswith("_par")])
    for c in range(n):
        data = z["r%d_data" % c]
        eps, m, xid, yid = z["r%d_par" % c].tolist()
        xl, got_xid = db.x_coordinate_clustering(data, eps, m)
        assert np.array_equal(xl, z["r%d_x" % c]) and got_xid == xid, c
        assert np.array_equal(db.main(data, eps, m), z["r%d_y" % c]), c


def test_dbscan_unsorted_x_golden(db, golden_dir):
    z = np.load(os.path.join(golden_dir, "dbscan_unsorted_x.npz"))
    n = len([k for k in z.files if k.endswith("_par")])
    for c in range(n):
        eps, m, xid = z["u%d_par" % c].tolist()
        xl, got = db.x_coordinate_clustering(z["u%d_data" % c], eps, m)
        assert np.array_equal(xl, z["u%d_x" % c]) and got == xid, c


def test_dbscan_gen_golden(db, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dbscan_gen.json")))
    for n in ("100000", "1000000", "5000000"):     # reference runs of 31 s, 5 min and 2 h 10 min
        if n not in g:
            continue
        lab = db.main(synth.gen_points(int(n)), g[n]["eps"], g[n]["m"])
        assert sha(lab.astype("<f8")) == g[n]["labels_sha256"], n
        assert int(lab.max()) == g[n]["final_max_id"] and int((lab == -1).sum()) == g[n]["n_noise"]


@pytest.mark.parametrize("n,span,eps,m,seed", [(200_000, 2_000_000, 500, 3, 1),      # dense: x-clusters of thousands of points
                                               (300_000, 100_000, 50, 3, 2),          # one giant x-cluster
                                               (150_000, 50_000_000, 500, 4, 3), (100_000, 5_000_000, 175, 2, 4),
                                               (50_000, 1_000, 5, 3, 5), (257, 300, 500, 3, 6), (100_000, 3_000_000, 300, 9, 7),
                                               (60_000, 400_000, 2000, 64, 8),      # largest m of the fused chained-scan path
                                               (60_000, 400_000, 2000, 65, 9), (5000, 20_000, 900, 200, 10),   # general multi-kernel path
                                               (1025, 4000, 50, 3, 11), (2048, 9000, 60, 2, 12), (1, 10, 5, 3, 13)])
def test_dbscan_dense_vs_oracle(db, n, span, eps, m, seed):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.integers(0, span, n))
    y = np.where(rng.random(n) < 0.5, x + rng.integers(0, 4 * eps, n), rng.integers(0, span, n))
    data = np.stack([x, y, np.arange(n)], 1).astype(np.int64)
    want = oracle.dbscan_main(data, eps, m)
    got = db.main(data, eps, m)
    assert np.array_equal(got, want)
    xl, xid = db.x_coordinate_clustering(data, eps, m)
    wl, wid = oracle.x_coordinate_clustering(data, eps, m)
    assert xid == wid and np.array_equal(xl, wl)


def test_dbscan_config3_sha(db):
    # BASELINE configs[2]: 5M points, e=500 l=3 — closed-form prediction recorded in SURVEY.md §8(d)
    pts = synth.gen_points(5_000_000)
    lab = db.main(pts, 500, 3)
    assert int(lab.max()) == 517346 and int((lab == -1).sum()) == 2283962
    assert sha(lab.astype("<f8")) == "2f3d430450eb79949107dae67154e5c6060e715e260e3547c537347ca1d0f45f"
    assert np.array_equal(lab, oracle.dbscan_main(pts, 500, 3))


def test_dbscan_many_tiles(db):
    # > 2048 tiles of 4096 points: tile prefixes come from the tile_scan launch instead of the in-kernel reduction
    pts = synth.gen_points(9_000_000, seed=5)
    assert np.array_equal(db.main(pts, 500, 3), oracle.dbscan_main(pts, 500, 3))


def test_dbscan_negative_and_offset_coordinates(db):
    rng = np.random.default_rng(3)
    x = np.sort(rng.integers(-50_000, 50_000, 5000)) + (1 << 40)
    y = rng.integers(-1000, 1000, 5000) - (1 << 35)
    data = np.stack([x, y], 1).astype(np.int64)
    assert np.array_equal(db.main(data, 40, 3), oracle.dbscan_main(data, 40, 3))
    assert np.array_equal(db.main(data, 39.5, 3), oracle.dbscan_main(data, 39.5, 3))


def _bucketed_case(rng, nb):
    sizes = rng.choice([0, 0, 1, 2, 3, 5, 40, 300, 5000], nb)
    xs, ys, want, lastid = [], [], [], []
    for s in sizes:
        x = np.sort(rng.integers(0, max(10, s * 40), s))
        y = np.where(rng.random(s) < 0.6, x + rng.integers(0, 900, s), rng.integers(0, max(10, s * 40), s))
        xs.append(x)
        ys.append(y)
        if s:
            d = np.stack([x, y], 1).astype(np.int64)
            xl, xid = oracle.x_coordinate_clustering(d, 300, 3)
            yl, yid = oracle.y_coordinate_clustering(d, 300, 3, xid, xl)
            want.append(yl)
            lastid.append(yid)
        else:
            want.append(np.zeros(0))
            lastid.append(-1)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return np.concatenate(xs), np.concatenate(ys), off, np.concatenate(want), np.array(lastid)


def test_dbscan_device_buckets(ctx, nat):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(21)
    for nb in (1, 7, 300):
        x, y, off, want, lastid = _bucketed_case(rng, nb)
        dev = torch.device("cuda:0")
        tx = torch.from_numpy(x.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
        ty = torch.from_numpy(y.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
        tl = torch.empty(max(1, len(x)), dtype=torch.float64, device=dev)
        tid = torch.empty(nb, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        nat.check(ctx.lib.tdt_dbscan_device(ctx.handle, tx.data_ptr(), ty.data_ptr(), len(x), nat.ptr(off), nb, 300, 3, 0,
                                            tl.data_ptr(), tid.data_ptr()))
        ctx.sync()
        assert np.array_equal(tl.cpu().numpy()[:len(x)], want), nb
        assert np.array_equal(tid.cpu().numpy(), lastid), nb


def test_sort_dbscan(ctx, nat):
    rng = np.random.default_rng(22)
    nb = 40
    x, y, off, _, _ = _bucketed_case(rng, nb)
    # scramble inside every bucket; duplicates in posA exercise the stable tie order
    x = (x // 7) * 7
    for b in range(nb):
        p = rng.permutation(off[b + 1] - off[b])
        x[off[b]:off[b + 1]] = x[off[b]:off[b + 1]][p]
        y[off[b]:off[b + 1]] = y[off[b]:off[b + 1]][p]
    n = len(x)
    perm = np.empty(n, dtype=np.uint32)
    lab = np.empty(n, dtype=np.float64)
    xa = x.astype(np.int64)
    ya = y.astype(np.int64)
    nat.check(ctx.lib.tdt_sort_dbscan(ctx.handle, nat.ptr(xa), nat.ptr(ya), n, nat.ptr(off), nb, 300.0, 3, nat.ptr(perm), nat.ptr(lab)))
    for b in range(nb):
        lo, hi = int(off[b]), int(off[b + 1])
        if lo == hi:
            continue
        order = np.argsort(xa[lo:hi], kind="stable")                     # sorted(..., key=posA)  tiddit_cluster.pyx:152
        assert np.array_equal(perm[lo:hi], lo + order), b
        d = np.stack([xa[lo:hi][order], ya[lo:hi][order]], 1)
        assert np.array_equal(lab[lo:hi], oracle.dbscan_main(d, 300, 3)), b


@pytest.mark.parametrize("n,nb,span", [(1, 1, 100), (2, 1, 1), (63, 1, 1 << 30), (4095, 1, 1 << 20), (4096, 3, 1 << 31), (4097, 1, 37),
                                       (12289, 300, 250_000_000), (200_001, 1, 1 << 31), (1_000_003, 300, 250_000_000), (1_500_000, 1, 5000),
                                       (300_000, 2, 1 << 9), (70_000, 129, 1 << 24)])
def test_sort_order_is_the_stable_order_at_every_size(ctx, nat, n, nb, span):
    """a12 (tiddit_cluster.pyx:152): the one-sweep radix sort behind tdt_sort_dbscan — tile boundaries (4096 pairs), one and many
    buckets (the bucket index shares a digit with the last position bits), position spans from one digit to 31 bits, heavy ties
    (their order is the arrival order), a constant column"""
    rng = np.random.default_rng(n * 31 + nb)
    cuts = np.sort(rng.integers(0, n + 1, nb - 1)) if nb > 1 else np.zeros(0, dtype=np.int64)
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    xa = rng.integers(0, span, n).astype(np.int64)
    if n > 1000:
        xa[rng.integers(0, n, n // 3)] = xa[0]                           # a third of the column is one value
    ya = rng.integers(0, 1 << 31, n).astype(np.int64)
    perm = np.empty(n, dtype=np.uint32)
    lab = np.empty(n, dtype=np.float64)
    nat.check(ctx.lib.tdt_sort_dbscan(ctx.handle, nat.ptr(xa), nat.ptr(ya), n, nat.ptr(off), nb, 300.0, 3, nat.ptr(perm), nat.ptr(lab)))
    ctx.sync()                                                           # (also: no look-back of the sort timed out)
    want = np.empty(n, dtype=np.int64)
    for b in range(nb):
        lo, hi = int(off[b]), int(off[b + 1])
        want[lo:hi] = lo + np.argsort(xa[lo:hi], kind="stable")
    assert np.array_equal(perm, want)


def test_cluster_columns_int32_signal_order(ctx, nat):
    """tdt_cluster_columns: int32 columns (pageable and pinned), labels in SIGNAL order, x-run counts and final ids per bucket — against
    stable argsort + the oracle per bucket; negative coordinates, a max_pos bound, one bucket and many, duplicates in posA"""
    from tiddit_amd import tiddit_cluster
    from tiddit_amd.hostutil import PinnedPool
    rng = np.random.default_rng(31)
    pool = PinnedPool()
    for nb, shift, bound in ((1, 0, 0), (40, 0, 0), (40, 0, 1), (7, -50_000, 0)):
        x, y, off, _, _ = _bucketed_case(rng, nb)
        x = (x // 5) * 5 + shift
        y = y + shift
        for b in range(nb):
            p = rng.permutation(off[b + 1] - off[b])
            x[off[b]:off[b + 1]] = x[off[b]:off[b + 1]][p]
            y[off[b]:off[b + 1]] = y[off[b]:off[b + 1]][p]
        n = len(x)
        for pinned in (False, True):
            if pinned:
                xa, ya, lab = pool.take("a", n, np.int32), pool.take("b", n, np.int32), pool.take("l", n, np.int32)
                xa[:], ya[:] = x, y
            else:
                xa, ya, lab = x.astype(np.int32), y.astype(np.int32), np.empty(n, dtype=np.int32)
            runs, last = np.zeros(nb, dtype=np.int64), np.zeros(nb, dtype=np.int64)
            mp = int(x.max()) if bound else 0
            nat.check(ctx.lib.tdt_cluster_columns(ctx.handle, nat.ptr(xa), nat.ptr(ya), n, nat.ptr(off), nb, 300.0, 3, mp, nat.ptr(lab),
                                                  nat.ptr(runs), nat.ptr(last)))
            for b in range(nb):
                lo, hi = int(off[b]), int(off[b + 1])
                if lo == hi:
                    assert runs[b] == 0 and last[b] == -1
                    continue
                order = np.argsort(x[lo:hi], kind="stable")
                d = np.stack([x[lo:hi][order], y[lo:hi][order]], 1).astype(np.int64)
                xl, xid = oracle.x_coordinate_clustering(d, 300, 3)
                yl, yid = oracle.y_coordinate_clustering(d, 300, 3, xid, xl.copy())
                want = np.empty(hi - lo)
                want[order] = yl
                assert np.array_equal(lab[lo:hi].astype(np.float64), want), (nb, shift, bound, pinned, b)
                assert runs[b] == xid + 1 and last[b] == yid
        # the Python entry point on the same buckets (int32 route) and on coordinates beyond int32 (int64 route)
        buckets = [np.stack([x[off[b]:off[b + 1]], y[off[b]:off[b + 1]]], 1).astype(np.int64) for b in range(nb)]
        a = tiddit_cluster.cluster_buckets(buckets, 300, 3)
        big = [bk + (1 << 33) for bk in buckets]
        c = tiddit_cluster.cluster_buckets(big, 300, 3)
        assert all(np.array_equal(u, v) for u, v in zip(a, c))
    pool.close()


# ------------------------------------------------------------------------------------- cluster
def _jsonable(c):
    if isinstance(c, dict):
        return {str(k): _jsonable(v) for k, v in c.items()}
    if isinstance(c, set):
        return {"__set__": sorted(_jsonable(x) for x in c)}
    if isinstance(c, (list, tuple)):
        return [_jsonable(x) for x in c]
    if isinstance(c, np.integer):
        return int(c)
    if isinstance(c, np.floating):
        return float(c)
    return c


def test_cluster_main_golden(ctx, golden_dir, tmp_path):
    from tiddit_amd import tiddit_cluster
    g = json.load(open(os.path.join(golden_dir, "cluster.json")))
    for t in g["find_discordant_pos"]:
        frag = ["q", "1", "1", "100", "250", t["revA"], "900", "1050", t["revB"]]
        assert list(tiddit_cluster.find_discordant_pos(frag, t["is_mp"])) == t["out"]
    for ci, case in enumerate(g["cases"]):
        d = tmp_path / ("case%d" % ci)
        (d / "p_tiddit").mkdir(parents=True)
        (d / "p_tiddit" / "discordants_S.tab").write_text("".join(l + "\n" for l in case["discordants_tab"]))
        (d / "p_tiddit" / "splits_S.tab").write_text("".join(l + "\n" for l in case["splits_tab"]))
        cand = tiddit_cluster.main(str(d / "p"), case["chromosomes"], case["contig_length"], ["S"], case["is_mp"], case["epsilon"],
                                   case["m"], case["max_ins_len"], case["min_contig"], True, case["min_reads"])
        # same keys, same values AND same insertion order (the order defines the VCF SV ids downstream)
        assert json.dumps(_jsonable(cand)) == json.dumps(case["candidates"]), ci


def test_coverage_multi_contig_single_launch(cov, ctx):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    contigs = [("a", 3_000_000), ("b", 1137), ("c", 1_500_001), ("d", 400), ("e", 2_000_000)]
    h = cov.CoverageHistogram(contigs, 500)
    items, keep, want = [], [], {}
    for i, (name, LN) in enumerate(contigs):
        if LN > 1000:
            s, e, mq, fl = synth.gen_reads(LN, 25, seed=50 + i)
            if name == "c":
                s, e, mq, fl = s[:-1], e[:-1], mq[:-1], fl[:-1]      # odd count: scalar tail path
        else:
            s, e, mq, fl = (np.zeros(0, np.int64),) * 2 + (np.zeros(0, np.uint8), np.zeros(0, np.uint16))
        want[name], _ = oracle.coverage_stream(s, e, mq, fl, LN, 500, 20)
        ts = [torch.from_numpy(s.astype(np.int32)).to(dev), torch.from_numpy(e.astype(np.int32)).to(dev),
              torch.from_numpy(mq).to(dev), torch.from_numpy(fl.view(np.int16)).to(dev)]
        keep.append(ts)
        items.append((name, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), len(s)))
    out = torch.zeros(h.total_bins(), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    h.push_device_multi(items, 20)
    h.finish_all_device(out.data_ptr())
    ctx.sync()
    res = out.cpu().numpy()
    for name, LN in contigs:
        o, nb = h.offset(name), h.nbins(name)[0]
        assert np.array_equal(res[o:o + nb], want[name]), name
        assert np.array_equal(h.finish(name), want[name]), name
    h.close()


# ------------------------------------------------------------------------------- signal select
def test_signal_select_vs_numpy(ctx, nat):
    rng = np.random.default_rng(31)
    for n in (0, 1, 63, 64, 4097, 300_000, 9_000_000):
        n_contigs = 6
        ok = np.array([1, 1, 0, 1, 0, 1], dtype=np.uint8)
        flag = rng.choice(np.array([0x63, 0x93, 0x1, 0x0, 0x403, 0x903, 0x103, 0x9, 0x5, 0x3], dtype=np.uint16), n)
        mapq = rng.integers(0, 61, n).astype(np.uint8)
        tid = rng.integers(-1, n_contigs, n).astype(np.int32)
        mate = np.where(rng.random(n) < 0.9, tid, rng.integers(-1, n_contigs, n)).astype(np.int32)
        tlen = rng.integers(-3000, 3000, n).astype(np.int32)
        min_q, max_ins = 20, 1000
        f = flag.astype(np.int64)
        okc = np.zeros(n, dtype=bool)
        okc[tid >= 0] = ok[tid[tid >= 0]] != 0
        want = np.flatnonzero(okc & ((f & 0x404) == 0) & ((f & 0x900) == 0) & (mapq >= min_q) & ((f & 0x8) == 0) & ((f & 0x1) != 0)
                              & (mate >= 0) & ((np.abs(tlen.astype(np.int64)) > max_ins) | (mate != tid)))
        out = np.empty(max(n, 1), dtype=np.uint32)
        cnt = ctypes.c_size_t(0)
        nat.check(ctx.lib.tdt_signal_select(ctx.handle, nat.ptr(flag), nat.ptr(mapq), nat.ptr(tid), nat.ptr(mate), nat.ptr(tlen), n,
                                            nat.ptr(ok), n_contigs, min_q, max_ins, nat.ptr(out), ctypes.byref(cnt)))
        assert cnt.value == len(want), n
        assert np.array_equal(out[:cnt.value], want), n


def test_cluster_buckets_sharded_single_rank_group(ctx):
    """the multi-GPU entry point with a 1-rank RCCL group: same labels as the local path"""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from tiddit_amd import tiddit_cluster
    rng = np.random.default_rng(8)
    buckets = []
    for s in (0, 50, 3000, 7, 800):
        x = rng.integers(0, max(10, s * 30), s)
        buckets.append(np.stack([x, x + rng.integers(0, 900, s), np.arange(s)], 1).astype(np.int64).reshape(s, 3))
    want = tiddit_cluster.cluster_buckets(buckets, 300, 3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        got = tiddit_cluster.cluster_buckets_sharded(buckets, 300, 3)
    finally:
        dist.destroy_process_group()
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    for b, w in zip(buckets, want):
        if len(b):
            d = b[np.argsort(b[:, 0], kind="stable")]
            lab = oracle.dbscan_main(d, 300, 3)
            back = np.empty(len(b))
            back[np.argsort(b[:, 0], kind="stable")] = lab
            assert np.array_equal(w, back)


def test_coverage_host_push_many_chunks(cov):
    """> 2 staging chunks of 4 M reads: exercises the pinned double-buffer ring and its event hand-over"""
    LN = 60_000_000
    start, end, mapq, flag = synth.gen_reads(LN, 25, seed=123)       # 10 M reads
    want, kept = oracle.coverage_stream(start, end, mapq, flag, LN, 50, 5)
    h = cov.CoverageHistogram([("c", LN)], 50)
    h.push("c", start, end, mapq, flag, 5)
    got = h.finish("c")
    assert h.kept() == kept and np.array_equal(got, want)
    h.close()


# --------------------------------------------------------------------------------------- ploidy
def test_masked_medians_vs_numpy(ctx):
    """determine_ploidy's masked medians (tiddit_coverage_analysis.pyx:14-27): numpy.median is the reference"""
    from tiddit_amd import tiddit_coverage_analysis as ca
    rng = np.random.default_rng(17)
    pairs = []
    for n in (0, 1, 2, 5, 1000, 250_001, 3_000_000, 7):
        cov = np.round(rng.gamma(9.0, 3.3, n), 3) * (rng.random(n) > 0.1)          # many ties, 10 % empty bins
        gc = np.where(rng.random(n) < 0.07, -1, rng.integers(0, 101, n)).astype(np.int8)
        pairs.append((cov, gc))
    pairs.append((np.zeros(50), np.zeros(50, np.int8)))                                  # nothing selected
    pairs.append((np.array([2.5, 2.5, 2.5, 7.0]), np.array([10, 10, -1, 10], np.int8)))
    med, overall = ca.masked_medians(pairs)
    allv = []
    for (cov, gc), m in zip(pairs, med):
        sel = cov[(cov > 0) & (gc != -1)]
        allv.append(sel)
        if len(sel):
            assert m == np.median(sel), len(cov)
        else:
            assert np.isnan(m)
    assert overall == np.median(np.concatenate(allv))


def test_determine_ploidy_table(ctx, tmp_path):
    from tiddit_amd import tiddit_coverage_analysis as ca
    rng = np.random.default_rng(2)
    cov = {"chr1": rng.gamma(20, 1.5, 40000), "chrX": rng.gamma(10, 1.5, 20000), "chrEmpty": np.zeros(100)}
    gc = {k: np.where(rng.random(len(v)) < 0.05, -1, 40).astype(np.int8) for k, v in cov.items()}
    lib = ca.determine_ploidy(cov, ["chr1", "chrX", "chrEmpty", "chrMissing"], {}, 2, str(tmp_path / "p"), None, "ref.fa", 50, {}, gc)
    # the reference's loop, literally
    want, allc = {}, []
    for ch in cov:
        tmp = [cov[ch][i] for i in range(len(cov[ch])) if cov[ch][i] > 0 and gc[ch][i] != -1]
        allc += tmp
        m = np.median(tmp) if tmp else np.nan
        want[ch] = 0 if np.isnan(m) else m
    assert lib["avg_coverage"] == np.median(allc)
    for ch in cov:
        assert lib["avg_coverage_" + ch] == want[ch]
        assert lib["contig_ploidy_" + ch] == int(round(2 * want[ch] / lib["avg_coverage"]))
    rows = open(str(tmp_path / "p.ploidies.tab")).read().splitlines()
    assert rows[0] == "Chromosome\tPloidy\tPloidy_rounded\tMean_coverage" and len(rows) == 4
    assert rows[1] == "chr1\t{}\t{}\t{}".format(want["chr1"] / lib["avg_coverage"] * 2, lib["contig_ploidy_chr1"], want["chr1"])


@pytest.mark.parametrize("width,eol", [(60, "\n"), (61, "\n"), (70, "\r\n"), (1, "\n"), (7, "\n"), (100000, "\n"), (50, "\n"), (64, "\n")])
def test_gc_from_fasta_layout_equals_stripped(tmp_path, width, eol):
    """binned_gc straight from the file bytes (line ends in place) == the oracle on the stripped sequence, for line widths that
    do and do not divide the bin, CRLF files, one-base lines, a single-line contig and contigs ending on / off a line end"""
    from tiddit_amd import tiddit_gc
    from tiddit_amd.fasta import FastaFile
    rng = np.random.default_rng(width)
    path = str(tmp_path / "r.fa")
    seqs = {}
    with open(path, "wb") as f:
        for name, ln in (("a", 100_003), ("b", 60 * 70), ("c", 49), ("d", 250_000), ("e", 1)):
            s = synth.gen_sequence(ln, seed=int(rng.integers(1, 1 << 30)))
            s[rng.integers(0, ln, max(1, ln // 50))] = ord("N")
            if ln > 5000:
                s[1000:1000 + 2600] = ord("n")
            seqs[name] = s
            f.write((">%s some description%s" % (name, eol)).encode())
            b = s.tobytes()
            for o in range(0, ln, width):
                f.write(b[o:o + width] + eol.encode())
    fa = FastaFile(path)
    for name, s in seqs.items():
        for z in (50, 500, 7, 2048, 3000):
            got = tiddit_gc.binned_gc(fa, name, z, 0.5)[1]
            assert np.array_equal(got, oracle.binned_gc(s, z, 0.5)), (name, z)
    g = tiddit_gc.main(path, ["a", "d"], 1, 50, 0.5)
    assert np.array_equal(g["a"], oracle.binned_gc(seqs["a"], 50, 0.5)) and np.array_equal(g["d"], oracle.binned_gc(seqs["d"], 50, 0.5))


def test_gc_of_many_small_contigs_in_groups(tmp_path, monkeypatch):
    """tiddit_gc.main (tiddit_gc.pyx:35-42) over a reference of many small contigs — the shape of GRCh38's alt / decoy / HLA contigs —
    goes to the device in GROUPS (tdt_gc_bins_fasta_many: one copy in, one launch per contig, one wait): contigs wrapped at different
    widths in one file, one-line and one-base contigs, an empty contig, a last contig without a final line end, groups of every size
    (the batch limit forced down), contigs above the grouping limit in between — all equal to the oracle per contig, keys in call order"""
    from tiddit_amd import tiddit_gc
    rng = np.random.default_rng(77)
    path = str(tmp_path / "many.fa")
    seqs, order = {}, []
    with open(path, "wb") as f:
        for i in range(420):
            name = ["HLA-A*%02d:%02d:01" % (1 + i % 40, 1 + i // 40), "chrUn_JTFH0100%04dv1_decoy" % i, "chr%d_KI27%04dv1_alt" % (1 + i % 22, i)][i % 3]
            ln = int(rng.choice([0, 1, 49, 50, 51, 60, 61, 970, 2274, int(rng.integers(100, 9000)), int(rng.integers(9000, 40000))])) if i else 33_000
            if i == 7:
                ln = 0
            width = int(rng.choice([60, 70, 80, 61]))
            s = synth.gen_sequence(max(ln, 1), seed=int(rng.integers(1, 1 << 30)))[:ln]
            if ln > 200:
                s[rng.integers(0, ln, ln // 40)] = ord("N")
                s[50:50 + min(130, ln - 60)] = ord("n")
            seqs[name] = s
            order.append(name)
            f.write((">%s\n" % name).encode())
            b = s.tobytes()
            last = i == 419
            for o in range(0, ln, width):
                f.write(b[o:o + width] + (b"" if last and o + width >= ln else b"\n"))
    assert len(seqs) == 420
    want = {n: oracle.binned_gc(seqs[n], 50, 0.5) if len(seqs[n]) else np.zeros(0, dtype=np.int8) for n in order}
    for batch, one in ((96 << 20, 8 << 20), (30_000, 8 << 20), (1, 8 << 20), (96 << 20, 5_000)):
        monkeypatch.setattr(tiddit_gc, "_MANY_MAX_BATCH", batch)
        monkeypatch.setattr(tiddit_gc, "_MANY_MAX_CONTIG", one)
        got = tiddit_gc.main(path, order[::-1], 1, 50, 0.5)
        assert list(got) == order[::-1]
        for n in order:
            assert got[n].dtype == np.int8 and np.array_equal(got[n], want[n]), (n, batch, one)
    got = tiddit_gc.main(path, order[:5], 1, 500, 0.5)             # another bin size through the same groups
    for n in order[:5]:
        assert np.array_equal(got[n], oracle.binned_gc(seqs[n], 500, 0.5) if len(seqs[n]) else np.zeros(0, dtype=np.int8))


def test_comm_abi_single_rank_group(ctx):
    """tdt_comm_* (RCCL bound at run time): a one-rank communicator on this GPU — the variable-count all-gather and the float64
    sum all-reduce are the identity; (more ranks need more GPUs: the N-rank layout logic is covered by the gloo tests)"""
    torch = pytest.importorskip("torch")
    from tiddit_amd.comm import Comm
    dev = torch.device("cuda:0")
    c = Comm(0, 1, Comm.unique_id(), ctx)
    src = torch.arange(1000, dtype=torch.int32, device=dev) * 3 - 7
    dst = torch.zeros(1000, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    c.allgatherv(src.data_ptr(), dst.data_ptr(), [1000], 4)
    bins = torch.rand(4096, dtype=torch.float64, device=dev)
    want = bins.clone()
    torch.cuda.synchronize()
    c.allreduce_sum_f64(bins.data_ptr(), bins.numel())
    ctx.sync()
    assert torch.equal(dst, src) and torch.equal(bins, want)
    empty = torch.zeros(1, dtype=torch.int32, device=dev)
    c.allgatherv(0, empty.data_ptr(), [0], 4)
    ctx.sync()
    c.close()


def test_segment_means_equal_numpy_average_bit_for_bit(ctx):
    """tiddit_variant.pyx:265-283,307-315: numpy.average of coverage slices / of the gc-masked slice.  The kernel reproduces
    numpy's chunked pairwise summation order, so the float64 means are identical, not merely close."""
    import warnings
    from tiddit_amd import tiddit_region
    rng = np.random.default_rng(3)
    cov = {"chr1": rng.gamma(30, 1.0, 2_500_000) * rng.choice([1e-3, 1.0, 1e3], 2_500_000), "chr2": rng.gamma(5, 2.0, 70_001), "tiny": np.zeros(3)}
    gc = {k: np.where(rng.random(len(v) + 2) < 0.15, -1, 41).astype(np.int8) for k, v in cov.items()}
    table = tiddit_region.BinTable(cov, gc)
    segs, masked = [], []
    for n in list(range(0, 140)) + [255, 256, 257, 1000, 8191, 8192, 8193, 16384, 16385, 20000, 100_001, 1 << 20, 2_400_000]:
        for off in (0, 1, 3, 5):
            segs.append(("chr1", off, off + n))
            masked.append(len(segs) % 2)
    segs += [("chr2", 69_990, 80_000), ("chr2", 500, 400), ("tiny", 0, 10), ("tiny", 5, 9), ("chr2", 0, 70_001)]
    masked += [0, 1, 1, 0, 1]
    mean, count = tiddit_region.region_means(table, segs, masked)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, (c, s, e) in enumerate(segs):
            a = cov[c][s:e]
            if masked[i]:
                a = a[gc[c][s:e][:len(a)] > -1]
            want = np.average(a) if len(a) else np.nan
            assert count[i] == len(a), (i, segs[i])
            assert (mean[i] == want) or (np.isnan(mean[i]) and np.isnan(want)), (i, segs[i], mean[i], want)
            o = oracle.np_masked_mean(cov[c][s:e], gc[c][s:e][:len(cov[c][s:e])])[0] if masked[i] else oracle.np_mean(cov[c][s:e])
            assert (o == want) or (np.isnan(o) and np.isnan(want))


def test_candidate_means_follow_define_variant(ctx):
    import math
    import warnings
    from tiddit_amd import tiddit_region
    rng = np.random.default_rng(8)
    cov = {"chr1": rng.gamma(30, 1.0, 40_000), "chr10": rng.gamma(30, 1.0, 9_000)}
    gc = {k: np.where(rng.random(len(v)) < 0.3, -1, 41).astype(np.int8) for k, v in cov.items()}
    lib = {"avg_coverage_chr1": 29.5, "avg_coverage_chr10": 31.0}
    cl = {"chr1": {"chr1": {}, "chr10": {}}}
    for k in range(60):
        a = int(rng.integers(1000, 1_900_000))
        b = a + int(rng.choice([120, 700, 1000, 1100, 5000, 90_000]))
        if k % 7 == 0:
            a, b = b, a
        cl["chr1"]["chr1"][k] = {"posA": a, "posB": b, "startA": min(a, b) - 300, "endA": min(a, b) + 5, "startB": max(a, b) - 2, "endB": max(a, b) + 310}
    cl["chr1"]["chr10"][3] = {"posA": 5000, "posB": 7000, "startA": 4700, "endA": 5001, "startB": 6990, "endB": 7300}
    cl["chr1"]["chr1"][99] = {"posA": 1_999_990, "posB": 2_100_000, "startA": 1_999_700, "endA": 2_000_400, "startB": 2_099_000, "endB": 2_100_000}   # past the contig end
    got = tiddit_region.candidate_means(cl, cov, gc, lib)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for chrA in cl:
            for chrB in cl[chrA]:
                for cid, c in cl[chrA][chrB].items():
                    posA, posB = c["posA"], c["posB"]
                    if chrA == chrB and posA > posB:
                        posA, posB = posB, posA
                    s, e = int(math.floor(c["startA"] / 50.0)), int(math.floor(c["endA"] / 50.0)) + 1
                    avg_a = np.average(cov[chrA][s:e])
                    s, e = int(math.floor(c["startB"] / 50.0)), int(math.floor(c["endB"] / 50.0)) + 1
                    avg_b = np.average(cov[chrB][s:e])
                    if chrA != chrB:
                        covM = 0
                    elif abs(posB - posA) < 1000:
                        covM = None
                    else:
                        s, e = int(math.floor(posA / 50.0)), int(math.floor(posB / 50.0)) + 1
                        between = cov[chrA][s:e][gc[chrA][s:e] > -1]
                        covM = np.average(between) if len(between) > 4 else lib["avg_coverage_" + chrA]
                    g = got[(chrA, chrB, cid)]
                    same = lambda x, y: (x is None and y is None) or x == y or (x is not None and y is not None and np.isnan(x) and np.isnan(y))
                    assert same(g["avg_a"], avg_a) and same(g["avg_b"], avg_b) and same(g["covM"], covM), (chrA, chrB, cid, g, avg_a, avg_b, covM)


@pytest.mark.parametrize("case", ["sparse", "deep", "long_sorted", "contig_end", "tiny_bins", "bin128", "forced_run_merged", "pile_up"])
def test_coverage_small_bin_flavour_corners(cov, case, monkeypatch):
    """the difference-pair flavour of cov_accumulate (bins <= 128 bp): streams that re-base the LDS window all the time, many
    reads per bin, sorted reads far longer than the register path takes, reads piled on the contig's last bins, 2-bp and
    128-bp bins; and the run-merged flavour forced onto 50-bp bins — all bit-identical to the per-read oracle"""
    rng = np.random.default_rng(["sparse", "deep", "long_sorted", "contig_end", "tiny_bins", "bin128", "forced_run_merged", "pile_up"].index(case) + 40)
    z, q = 50, 5
    if case == "sparse":            # 0.2x: a tile of 1024 reads spans far more than the window
        LN, n = 40_000_000, 60_000
        start = np.sort(rng.integers(0, LN - 400, n))
        span = rng.integers(1, 300, n)
    elif case == "deep":            # 3000x on a short contig: hundreds of reads per bin
        LN, n = 200_000, 4_000_000
        start = np.sort(rng.integers(0, LN - 160, n))
        span = rng.integers(100, 160, n)
    elif case == "long_sorted":     # sorted long reads: 5 % beyond 256 bins, some beyond the whole window
        LN, n = 30_000_000, 300_000
        start = np.sort(rng.integers(0, LN - 1, n))
        span = np.where(rng.random(n) < 0.05, rng.integers(12_000, 400_000, n), rng.integers(50, 12_900, n))
    elif case == "contig_end":      # everything within the last few hundred bins, contig length not a multiple of the bin
        LN, n = 1_000_037, 500_000
        start = np.sort(rng.integers(LN - 20_000, LN - 1, n))
        span = rng.integers(1, 3000, n)
    elif case == "pile_up":         # whole workgroups (16 384 reads) of IDENTICAL reads: every +1 of a window on one word, every -1 on another —
        LN = 2_000_000              # the difference counts at the edge of the 16-bit fields the window resolve packs and scans
        pos = np.repeat(np.array([1_000, 1_049, 7_777, 500_025, 500_026, 1_999_700]), [16_384 * 3, 16_384, 40_000, 16_384 * 2 + 5, 16_383, 20_000])
        span = np.repeat(np.array([150, 101, 12_000, 149, 51, 299]), [16_384 * 3, 16_384, 40_000, 16_384 * 2 + 5, 16_383, 20_000])
        n, start = len(pos), pos
    elif case == "tiny_bins":
        z, LN, n = 2, 300_001, 400_000
        start = np.sort(rng.integers(0, LN - 1, n))
        span = rng.integers(1, 700, n)
    elif case == "bin128":
        z, LN, n = 128, 9_000_001, 1_500_000
        start = np.sort(rng.integers(0, LN - 1, n))
        span = rng.integers(1, 500, n)
    else:
        monkeypatch.setenv("TIDDIT_COV_MODE", "0")
        LN, n = 3_000_000, 600_000
        start = np.sort(rng.integers(0, LN - 200, n))
        span = rng.integers(1, 200, n)
    end = np.minimum(start + span, LN)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    flag = np.where(rng.random(n) < 0.05, 0x400, 0).astype(np.uint16) | np.where(rng.random(n) < 0.02, 0x4, 0).astype(np.uint16)
    if case == "pile_up":           # (nothing filtered: the counts must reach 2^14)
        mapq[:] = 60
        flag[:] = 0
    want, kept = oracle.coverage_stream(start, end, mapq, flag, LN, z, q)
    h = cov.CoverageHistogram([("c", LN)], z)
    h.push("c", start, end, mapq, flag, q)
    got = h.finish("c")
    assert h.kept() == kept
    h.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("z,q", [(500, 20), (50, 5), (1, 0), (4096, 30)])
def test_coverage_packed_records_equal_the_four_arrays(cov, ctx, z, q):
    """8-byte packed records (start | span:24 mapq:6 unmapped dup): same bins and kept count as the start/end/mapq/flag arrays,
    including reads >= 16 Mb that escape to the end array, mapq above 63 and every flag combination"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    contigs = [("a", 40_000_000), ("b", 1137), ("c", 2_000_003)]
    h = cov.CoverageHistogram(contigs, z)
    hp = cov.CoverageHistogram(contigs, z)
    items, pitems, keep, want = [], [], [], {}
    rng = np.random.default_rng(z)
    for i, (name, LN) in enumerate(contigs):
        s, e, mq, fl = synth.gen_reads(LN, 20 if LN > 2000 else 200, seed=70 + i)
        mq = np.where(rng.random(len(mq)) < 0.05, rng.integers(64, 256, len(mq)), mq).astype(np.uint8)      # above the 6-bit field
        if name == "a":                                                                                    # two reads of >= 16 Mb
            e[5], e[len(e) // 2] = s[5] + 17_000_000, min(LN, s[len(e) // 2] + 16_777_215)
        want[name], _ = oracle.coverage_stream(s, e, mq, fl, LN, z, q)
        ts = [torch.from_numpy(s.astype(np.int32)).to(dev), torch.from_numpy(e.astype(np.int32)).to(dev),
              torch.from_numpy(mq).to(dev), torch.from_numpy(fl.view(np.int16)).to(dev)]
        pk = torch.empty(len(s), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        nat_check = __import__("tiddit_amd._native", fromlist=["check"]).check
        nat_check(ctx.lib.tdt_cov_pack_device(ctx.handle, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), len(s), pk.data_ptr()))
        keep += ts + [pk]
        items.append((name, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), len(s)))
        pitems.append((name, pk.data_ptr(), ts[1].data_ptr(), len(s)))
    h.push_device_multi(items, q)
    hp.push_packed_device_multi(pitems, q)
    for name, _ in contigs:
        a, b = h.finish(name), hp.finish(name)
        assert np.array_equal(a, want[name]) and np.array_equal(b, want[name]), name
    assert h.kept() == hp.kept()
    h.close()
    hp.close()


def test_y_pass_on_caller_supplied_labels(db):
    """DBSCAN.y_coordinate_clustering takes any label array of the x pass's shape — here the x labels of ANOTHER eps / m, labels with
    some clusters dropped and the rest renumbered by hand, and an arbitrary starting cluster_id — not only the ones this module produced"""
    rng = np.random.default_rng(21)
    for n, span, eps_x, m_x, eps, m, start_id in ((40_000, 30_000_000, 800, 3, 300, 3, None), (25_000, 9_000_000, 500, 4, 500, 2, 777),
                                                  (3000, 400_000, 400, 3, 150, 4, 5), (64, 4000, 300, 2, 300, 3, None)):
        x = np.sort(rng.integers(0, span, n))
        y = np.where(rng.random(n) < 0.6, x + rng.integers(0, 6 * eps, n), rng.integers(0, span, n))
        data = np.stack([x, y, np.arange(n)], 1).astype(np.int64)
        xl, xid = oracle.x_coordinate_clustering(data, eps_x, m_x)
        if start_id is not None and xid >= 3:                       # drop every third cluster, renumber the rest 0, 1, 2, ...
            ids = xl.astype(np.int64)
            kept = (ids >= 0) & (ids % 3 != 2)
            new_id = np.cumsum(np.arange(xid + 1) % 3 != 2) - 1
            xl = np.where(kept, new_id[np.maximum(ids, 0)], -1).astype(np.float64)
            xid = int(xl.max())
        cid = xid if start_id is None else start_id
        want, wid = oracle.y_coordinate_clustering_labels(data, eps, m, cid, xl.copy())       # the literal restatement (pinned to DBSCAN.py)
        mine = xl.copy()
        got, gid = db.y_coordinate_clustering(data, eps, m, cid, mine)
        assert got is mine and gid == wid and np.array_equal(got, want), (n, eps_x, m_x, eps, m)
    # the same label on two separate ranges: members are selected by value (DBSCAN.py:72), so this is one cluster of three
    lab = np.array([0.0, 0.0, -1.0, 0.0, 1.0, 1.0])
    d6 = np.stack([np.arange(6) * 10, np.arange(6), np.arange(6)], 1).astype(np.int64)
    want, wid = oracle.y_coordinate_clustering_labels(d6, 50, 2, 1, lab)
    got, gid = db.y_coordinate_clustering(d6, 50, 2, 1, lab.copy())
    assert gid == wid and np.array_equal(got, want)


def test_y_pass_on_arbitrary_labels_equals_the_reference(db, golden_dir):
    """DBSCAN.y_coordinate_clustering on every label shape the reference accepts (DBSCAN.py:66-123), against vectors made with the REAL
    DBSCAN.py (tests/golden/dbscan_y_labels.npz): labels of another eps / m, clusters far above 128 members, a value on several
    ranges, sparse / float / negative values, 3000 distinct values (the visiting order of set() matters), and a cluster_id below the
    largest label, where produced ids collide with clusters not visited yet — no NotImplementedError left"""
    z = np.load(os.path.join(golden_dir, "dbscan_y_labels.npz"))
    n = sum(1 for k in z.files if k.endswith("_par"))
    assert n >= 120
    for c in range(n):
        eps, m, cid, want_id = (int(v) for v in z["c%d_par" % c])
        mine = z["c%d_in" % c].copy()
        got, gid = db.y_coordinate_clustering(z["c%d_data" % c], eps, m, cid, mine)
        assert got is mine and gid == want_id and np.array_equal(got, z["c%d_out" % c]), (c, str(z["c%d_kind" % c]))


def test_y_pass_large_and_colliding_labels_against_the_literal_restatement(db):
    """sizes the golden vectors do not reach: one 30 000-member cluster next to small ones (the old 128-member limit), 2 000 clusters
    with cluster_id below the largest label (the literal replay), float labels; vs oracle.y_coordinate_clustering_labels"""
    rng = np.random.default_rng(77)
    x = np.sort(rng.integers(0, 3_000_000, 40_000))
    y = x + rng.integers(0, 2000, len(x))
    data = np.stack([x, y], 1).astype(np.int64)
    lab = np.full(len(x), -1.0)
    lab[1000:31_000] = 0
    lab[31_500:31_600] = 1
    lab[32_000:32_050] = 2.5
    for cid in (2, 3, 10):
        want, wid = oracle.y_coordinate_clustering_labels(data, 40, 3, cid, lab)
        got, gid = db.y_coordinate_clustering(data, 40, 3, cid, lab.copy())
        assert gid == wid and np.array_equal(got, want), cid
    n = 6000
    x = np.sort(rng.integers(0, 800_000, n))
    data = np.stack([x, x + rng.integers(0, 300, n)], 1).astype(np.int64)
    lab = np.sort(rng.integers(-1, 2000, n)).astype(np.float64)
    for cid in (-1, 50, 1000, 1999):
        want, wid = oracle.y_coordinate_clustering_labels(data, 60, 2, cid, lab)
        got, gid = db.y_coordinate_clustering(data, 60, 2, cid, lab.copy())
        assert gid == wid and np.array_equal(got, want), cid
    with pytest.raises(ValueError):                               # m = 1: max() of an empty window, like the reference
        db.y_coordinate_clustering(data, 60, 1, 5, lab.copy())


def test_coverage_config2_whole_genome_every_contig(cov, ctx):
    """BASELINE configs[1] at its full size — 24 contigs x 125 Mb, 30x, 600 M reads in ONE launch — both as `--cov` runs it
    (500-bp bins, q >= 20) and as `--sv` does (50-bp bins, q >= 5; tiddit_signal.pyx:181,235), from the packed records and from the
    four arrays: every bin of every contig and the kept count equal the scalar oracle's"""
    torch = pytest.importorskip("torch")
    from tiddit_amd import _native
    dev = torch.device("cuda:0")
    C, L = 24, 125_000_000
    if torch.cuda.get_device_properties(0).total_memory < 40 * 2**30:
        pytest.skip("needs the stream of a 3-Gb genome in device memory")
    reads = [synth.gen_reads_device(L, 30, dev, seed=synth.SEED + c) for c in range(C)]
    torch.cuda.synchronize()
    n = [int(r[0].numel()) for r in reads]
    assert sum(n) == 600_000_000
    packed = []
    for c in range(C):
        pk = torch.empty(n[c], dtype=torch.int64, device=dev)
        _native.check(ctx.lib.tdt_cov_pack_device(ctx.handle, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(),
                                                  reads[c][3].data_ptr(), n[c], pk.data_ptr()))
        packed.append(pk)
    ctx.sync()
    host = [tuple(t.cpu().numpy() for t in r) for r in reads]
    contigs = [("s%02d" % (c + 1), L) for c in range(C)]
    for z, q in ((500, 20), (50, 5)):
        want, kept = [], 0
        for c in range(C):
            s, e, mq, fl = host[c]
            w, k = oracle.coverage_stream(s, e, mq, fl.view(np.uint16), L, z, q)
            want.append(w)
            kept += k
        for layout in ("binned", "packed", "four arrays"):
            h = cov.CoverageHistogram(contigs, z)
            if layout == "binned":                      # records made for THIS bin size (what the bound ingest kernel writes)
                bn = torch.empty(sum(n), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                offs = np.concatenate([[0], np.cumsum(n)])
                for c in range(C):
                    h.pack_binned_device(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c],
                                         bn.data_ptr() + 8 * int(offs[c]))
                h.push_binned_device_multi([(c, bn.data_ptr() + 8 * int(offs[c]), reads[c][0].data_ptr(), reads[c][1].data_ptr(), n[c]) for c in range(C)], q)
            elif layout == "packed":
                h.push_packed_device_multi([(c, packed[c].data_ptr(), reads[c][1].data_ptr(), n[c]) for c in range(C)], q)
            else:
                h.push_device_multi([(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c])
                                     for c in range(C)], q)
            for c in range(C):
                assert np.array_equal(h.finish(contigs[c][0]), want[c]), (z, layout, c)
            assert h.kept() == kept, (z, layout)
            h.close()
            if layout == "binned":
                del bn


@pytest.mark.parametrize("seed", range(8))
def test_coverage_random_parameter_sweep(cov, ctx, seed):
    """Random bin sizes on both sides of every kernel switch (1, the table-driven 2..1023 with their 24-bit division constants, the
    difference-pair flavour <= 128, >= 1024), contig lengths that are and are not multiples of the bin, short / long / mixed reads,
    sorted and shuffled streams, several filters — binned records, packed records and the four arrays against the scalar oracle"""
    torch = pytest.importorskip("torch")
    from tiddit_amd import _native
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        z = int(rng.choice([1, 2, 3, 7, 31, 50, 64, 100, 127, 128, 129, 250, 500, 511, 777, 1000, 1023, 1024, 1500, 5000, int(rng.integers(2, 1400))]))
        LN = int(rng.integers(1, 3_000_000)) if rng.random() < 0.8 else int(z * rng.integers(1, 2000))
        n = int(rng.integers(1, 400_000))
        start = np.sort(rng.integers(0, LN, n))
        kind = rng.random()
        if kind < 0.4:
            span = rng.integers(1, 300, n)                                  # short reads
        elif kind < 0.7:
            span = np.where(rng.random(n) < 0.03, rng.integers(1, 100_000, n), rng.integers(1, 400, n))      # long-read tails
        else:
            span = rng.integers(1, max(2, 6 * z), n)                        # a few bins each, whatever the bin size
        end = np.minimum(start + span, LN)
        if rng.random() < 0.3:
            p = rng.permutation(n)                                          # unsorted input: any order is legal for the histogram
            start, end = start[p], end[p]
        mapq = rng.choice([0, 1, 4, 5, 19, 20, 30, 60, 61, 255], n).astype(np.uint8)
        flag = (rng.choice([0, 0x4, 0x400, 0x404, 0x10, 0x800], n, p=[.8, .04, .04, .02, .05, .05]) | rng.choice([0, 0x1, 0x2, 0x100], n)).astype(np.uint16)
        q = int(rng.choice([0, 1, 5, 20, 60, 63]))
        s32, e32 = start.astype(np.int32), end.astype(np.int32)
        want, kept = oracle.coverage_stream(s32, e32, mapq, flag, LN, z, q)
        ts = [torch.from_numpy(s32).to(dev), torch.from_numpy(e32).to(dev), torch.from_numpy(mapq).to(dev),
              torch.from_numpy(flag.view(np.int16)).to(dev)]
        pk = torch.empty(n, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        _native.check(ctx.lib.tdt_cov_pack_device(ctx.handle, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n, pk.data_ptr()))
        for layout in ("binned", "packed", "four arrays"):
            h = cov.CoverageHistogram([("c", LN)], z)
            if layout == "binned":
                bn = torch.empty(n, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                if not h.has_binned():                  # bin sizes 1 and >= 1024 have no binned form: refused, not silently handled
                    with pytest.raises(_native.TdtError):
                        h.pack_binned_device("c", ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n, bn.data_ptr())
                    h.close()
                    continue
                h.pack_binned_device("c", ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n, bn.data_ptr())
                h.push_binned_device_multi([("c", bn.data_ptr(), ts[0].data_ptr(), ts[1].data_ptr(), n)], q)
            elif layout == "packed":
                h.push_packed_device_multi([("c", pk.data_ptr(), ts[1].data_ptr(), n)], q)
            else:
                h.push_device_multi([("c", ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n)], q)
            got = h.finish("c")
            assert h.kept() == kept, (seed, z, LN, n, q, layout)
            assert np.array_equal(got, want), (seed, z, LN, n, q, layout)
            h.close()


@pytest.mark.parametrize("seed", range(6))
def test_dbscan_random_parameter_sweep(ctx, nat, seed):
    """Random (eps, m) with m on both sides of the m == 3 fast path and up to the tile path's limit, sorted and unsorted x, dense and
    sparse buckets, 1 .. 60 buckets in one call, both modes (x pass only / both passes): labels and last ids against the oracle"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2000 + seed)
    for _ in range(5):
        nb = int(rng.choice([1, 1, 2, 9, 60]))
        m = int(rng.choice([2, 2, 3, 3, 4, 5, 8, 17, 64]))          # (m = 1: the reference itself raises, max() of an empty window)
        eps = int(rng.choice([1, 50, 300, 500, 5000]))
        mode = int(rng.random() < 0.25)
        srt = rng.random() < 0.7
        xs, ys, want, lastid = [], [], [], []
        sizes = rng.choice([0, 1, 2, m, m + 1, 70, 1500, 9000], nb)
        for s in sizes:
            s = int(s)
            span = max(10, int(s * eps * rng.choice([0.05, 0.5, 3.0])))
            x = rng.integers(0, span, s)
            if srt:
                x = np.sort(x)
            y = np.where(rng.random(s) < 0.6, x + rng.integers(0, 3 * eps + 1, s), rng.integers(0, span, s))
            xs.append(x)
            ys.append(y)
            if s:
                d = np.stack([x, y], 1).astype(np.int64)
                xl, xid = oracle.x_coordinate_clustering(d, eps, m)
                if mode == 1:
                    want.append(xl)
                    lastid.append(xid)
                else:
                    yl, yid = oracle.y_coordinate_clustering(d, eps, m, xid, xl)
                    want.append(yl)
                    lastid.append(yid)
            else:
                want.append(np.zeros(0))
                lastid.append(-1)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        x, y, want = np.concatenate(xs), np.concatenate(ys), np.concatenate(want)
        tx = torch.from_numpy(x.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
        ty = torch.from_numpy(y.astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
        tl = torch.empty(max(1, len(x)), dtype=torch.float64, device=dev)
        tid = torch.empty(nb, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        nat.check(ctx.lib.tdt_dbscan_device(ctx.handle, tx.data_ptr(), ty.data_ptr(), len(x), nat.ptr(off), nb, eps, m, mode,
                                            tl.data_ptr(), tid.data_ptr()))
        ctx.sync()
        key = (seed, nb, m, eps, mode, bool(srt), sizes.tolist())
        assert np.array_equal(tl.cpu().numpy()[:len(x)], want), key
        assert np.array_equal(tid.cpu().numpy(), np.array(lastid)), key


def _native_mod():
    from tiddit_amd import _native
    return _native


def test_binned_records_fields_and_edge_reads(cov, ctx):
    """cov_bin_record (csrc/tdt_common.h) field by field against tiddit_coverage.pyx:50-63 computed in numpy, and the reads that must
    leave the register path: three or more bins (500-bp flavour), a last bin that is the contig's last (other denominator, :67-69),
    single-bin reads IN the contig's last bin (which do use bin_size, :53-57), invalid reads (IndexError when they pass the filter)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    for z, LN in ((500, 10_250), (50, 10_030), (128, 9_000), (129, 9_000), (2, 101), (1023, 40_000)):
        rng = np.random.default_rng(z)
        n = 20_000
        start = np.sort(rng.integers(0, LN, n))
        span = np.where(rng.random(n) < 0.5, rng.integers(1, 2 * z + 2, n), rng.integers(1, 8 * z, n))
        end = np.minimum(start + span, LN)
        mapq = rng.choice([0, 5, 20, 60, 255], n).astype(np.uint8)
        flag = rng.choice([0, 0x4, 0x400, 0x10], n, p=[.85, .05, .05, .05]).astype(np.uint16)
        s32, e32 = start.astype(np.int32), end.astype(np.int32)
        ts = [torch.from_numpy(a).to(dev) for a in (s32, e32, mapq, flag.view(np.int16))]
        h = cov.CoverageHistogram([("c", LN)], z)
        bn = torch.empty(n, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        h.pack_binned_device("c", ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), n, bn.data_ptr())
        ctx.sync()
        rec = bn.cpu().numpy().view(np.uint64)
        lo, hi = (rec & 0xffffffff).astype(np.int64), (rec >> 32).astype(np.int64)
        nb = -(-LN // z)
        fb, eb = start // z, (end - 1) // z
        small = z <= 128
        shape = np.where(eb == fb, 0, np.where((eb >= nb - 1) | (eb - fb > (255 if small else 1)), 2, 1))
        assert np.array_equal(lo >> 2, fb) and np.array_equal(lo & 3, shape)
        assert np.array_equal(hi >> 30, ((flag & 0x400) != 0) * 2 + ((flag & 0x4) != 0)) and np.array_equal((hi >> 24) & 63, np.minimum(mapq, 63))
        bf = np.where(shape == 0, end - start, np.where(shape == 1, (fb + 1) * z - start, 0))
        bl = np.where(shape == 1, (end - 1) - eb * z, 0)
        if small:
            assert np.array_equal(hi & 0xff, bf) and np.array_equal((hi >> 8) & 0xff, bl)
        else:                                       # the first-bin index is stored scaled to the 16-byte table entries
            assert np.array_equal(hi & 0x3fff, bf << 4) and np.array_equal((hi >> 14) & 0x3ff, bl)
        if small:
            assert np.array_equal((hi >> 16) & 0xff, np.where(shape == 1, eb - fb, 0))
        assert (shape == 2).sum() > 50 and ((shape == 0) & (fb == nb - 1)).sum() > 0
        for q in (0, 20):
            h.reset()
            h.push_binned_device_multi([("c", bn.data_ptr(), ts[0].data_ptr(), ts[1].data_ptr(), n)], q)
            want, kept = oracle.coverage_stream(s32, e32, mapq, flag, LN, z, q)
            assert np.array_equal(h.finish("c"), want) and h.kept() == kept, (z, q)
        # a read that ends beyond the contig, and one with end <= start: IndexError if (and only if) it passes the filter
        for bad_end, bad_flag, raises in ((LN + 3 * z, 0, True), (int(s32[5]), 0, True), (LN + 3 * z, 0x400, False)):
            e2, f2 = e32.copy(), flag.copy()
            e2[5], f2[5] = bad_end, bad_flag
            t1, t3 = torch.from_numpy(e2).to(dev), torch.from_numpy(f2.view(np.int16)).to(dev)
            torch.cuda.synchronize()
            h.reset()
            h.pack_binned_device("c", ts[0].data_ptr(), t1.data_ptr(), ts[2].data_ptr(), t3.data_ptr(), n, bn.data_ptr())
            h.push_binned_device_multi([("c", bn.data_ptr(), ts[0].data_ptr(), t1.data_ptr(), n)], 0)
            if raises:
                with pytest.raises(_native_mod().TdtError):
                    h.finish("c")
            else:
                h.finish("c")
        h.close()


def test_stream_read_yardstick_runs_and_rejects_bad_arguments():
    """tdt_calib_stream_read (bench.py's roofline.stream_read): both walks return a positive time for a 1-GB buffer (larger than the
    256-MB last-level cache) and a rate that is at least plausible for HBM (> 1 TB/s, below the data sheet's 8 with some slack for the
    cached part); bad arguments are errors, not crashes"""
    import torch
    from tiddit_amd import _native
    ctx = _native.default_context()
    buf = torch.zeros(128 << 20, dtype=torch.int64, device="cuda")          # 1 GB
    torch.cuda.synchronize()
    for wpc, blocked in ((2, 0), (8, 1)):
        best, mean = ctypes.c_double(0), ctypes.c_double(0)
        _native.check(ctx.lib.tdt_calib_stream_read(ctx.handle, buf.data_ptr(), buf.numel() * 8, 5, wpc, blocked, ctypes.byref(best), ctypes.byref(mean)))
        assert 0 < best.value <= mean.value
        assert 1e12 < buf.numel() * 8 / (best.value * 1e-3) < 10e12
    best, mean = ctypes.c_double(0), ctypes.c_double(0)
    assert ctx.lib.tdt_calib_stream_read(ctx.handle, buf.data_ptr() + 8, buf.numel() * 8 - 8, 5, 8, 0, ctypes.byref(best), ctypes.byref(mean)) != 0   # misaligned
    assert ctx.lib.tdt_calib_stream_read(ctx.handle, buf.data_ptr(), 1024, 5, 8, 0, ctypes.byref(best), ctypes.byref(mean)) != 0                      # too small
    assert ctx.lib.tdt_calib_stream_read(ctx.handle, buf.data_ptr(), buf.numel() * 8, 5, 0, 0, ctypes.byref(best), ctypes.byref(mean)) != 0            # no workgroups
