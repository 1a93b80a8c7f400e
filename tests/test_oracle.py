"""Pin the CPU oracle (oracle/tiddit_oracle.c) to the golden vectors captured from the real
reference (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from tiddit_amd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def cov_golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "coverage.json")))


def test_create_coverage_shapes(cov_golden):
    for s in cov_golden["shapes"]:
        arr, ebs = oracle.create_coverage(s["LN"], s["bin"])
        assert (len(arr), ebs) == (s["nbins"], s["end_bin_size"])
        arr, ebs = oracle.create_coverage(77, s["bin"])
        assert (len(arr), ebs) == (s["nbins_b"], s["end_bin_size_b"])


def test_update_coverage_kat(cov_golden):
    for k in cov_golden["kat"]:
        arr, ebs = oracle.create_coverage(k["LN"], k["bin"])
        oracle.update_coverage(k["s"], k["e"], k["bin"], arr, ebs)
        assert arr.tolist() == k["bins"], k


def test_update_coverage_random(golden_dir):
    z = np.load(os.path.join(golden_dir, "coverage_random.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    assert n >= 10
    for c in range(n):
        LN, b = z["c%d_meta" % c].tolist()
        arr, ebs = oracle.create_coverage(LN, b)
        for s, e in zip(z["c%d_start" % c].tolist(), z["c%d_end" % c].tolist()):
            oracle.update_coverage(s, e, b, arr, ebs)
        assert np.array_equal(arr, z["c%d_bins" % c]), c


def test_update_coverage_out_of_range():
    arr, ebs = oracle.create_coverage(1000, 500)
    with pytest.raises(IndexError):
        oracle.update_coverage(900, 1200, 500, arr, ebs)


@pytest.mark.parametrize("tag", ["cov", "sv"])
def test_config1_stream(cov_golden, golden_dir, tag):
    g = cov_golden["config1"][tag]
    start, end, mapq, flag = synth.gen_reads(1_000_000, 10)
    bins, kept = oracle.coverage_stream(start, end, mapq, flag, 1_000_000, g["bin"], g["q"])
    assert kept == g["kept"] and len(bins) == g["nbins"]
    assert sha(bins.astype("<f8")) == g["bins_sha256"]
    assert np.array_equal(bins, np.load(os.path.join(golden_dir, "config1_bins_%s.npy" % tag)))
    assert float(bins.sum()) == g["sum"]


def test_gc_kat(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "gc.json")))
    for k in g["kat"]:
        out = oracle.binned_gc(k["seq"].encode(), k["bin"], k["n_cutoff"])
        assert out.tolist() == k["out"], k


def test_gc_random(golden_dir):
    z = np.load(os.path.join(golden_dir, "gc_random.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        L, b, seed = z["c%d_meta" % c].tolist()
        seq = synth.gen_sequence(L, seed=seed, n_frac=0.08)
        out = oracle.binned_gc(seq, b, float(z["c%d_cut" % c][0]))
        assert np.array_equal(out, z["c%d_out" % c]), c


def test_dbscan_kat(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dbscan.json")))
    for k in g["kat"]:
        data = np.array(k["data"], dtype=np.int64)
        for literal in (False, True):
            xl, xid = oracle.x_coordinate_clustering(data, k["eps"], k["m"])
            assert xl.tolist() == k["x"] and xid == k["x_id"], k
            yl, yid = oracle.y_coordinate_clustering(data, k["eps"], k["m"], xid, xl.copy(), literal=literal)
            assert yl.tolist() == k["y"] and yid == k["y_id"], k
            assert oracle.dbscan_main(data, k["eps"], k["m"], literal=literal).tolist() == k["y"]


def test_dbscan_random(golden_dir):
    z = np.load(os.path.join(golden_dir, "dbscan_random.npz"))
    n = len([k for k in z.files if k.endswith("_par")])
    assert n > 200
    for c in range(n):
        data = z["r%d_data" % c]
        eps, m, xid, yid = z["r%d_par" % c].tolist()
        xl, got_xid = oracle.x_coordinate_clustering(data, eps, m)
        assert np.array_equal(xl, z["r%d_x" % c]) and got_xid == xid, c
        for literal in (False, True):
            yl, got_yid = oracle.y_coordinate_clustering(data, eps, m, xid, xl.copy(), literal=literal)
            assert np.array_equal(yl, z["r%d_y" % c]) and got_yid == yid, (c, literal)


def test_dbscan_unsorted_x(golden_dir):
    z = np.load(os.path.join(golden_dir, "dbscan_unsorted_x.npz"))
    n = len([k for k in z.files if k.endswith("_par")])
    for c in range(n):
        eps, m, xid = z["u%d_par" % c].tolist()
        xl, got = oracle.x_coordinate_clustering(z["u%d_data" % c], eps, m)
        assert np.array_equal(xl, z["u%d_x" % c]) and got == xid, c


def test_dbscan_gen_100k(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dbscan_gen.json")))["100000"]
    pts = synth.gen_points(100_000)
    lab = oracle.dbscan_main(pts, g["eps"], g["m"])
    assert sha(lab.astype("<f8")) == g["labels_sha256"]
    assert np.array_equal(lab.astype(np.int32), np.load(os.path.join(golden_dir, "dbscan_100k_labels_i32.npy")))
    assert int(lab.max()) == g["final_max_id"] and int((lab == -1).sum()) == g["n_noise"]
    lit = oracle.dbscan_main(pts[:20000], g["eps"], g["m"], literal=True)
    assert np.array_equal(lit, oracle.dbscan_main(pts[:20000], g["eps"], g["m"]))


def test_dbscan_gen_1m(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "dbscan_gen.json"))).get("1000000")
    if g is None:
        pytest.skip("1M reference labels not generated (make_golden.py --slow)")
    lab = oracle.dbscan_main(synth.gen_points(1_000_000), g["eps"], g["m"])
    assert sha(lab.astype("<f8")) == g["labels_sha256"]


def test_np_mean_restatement_equals_numpy():
    """orc_np_mean / orc_np_masked_mean (numpy's chunked pairwise summation, tiddit_variant.pyx:265-283,307-315) against numpy itself"""
    import warnings
    rng = np.random.default_rng(1)
    for n in list(range(0, 140)) + [255, 256, 257, 1000, 8191, 8192, 8193, 9000, 16384, 16385, 20000, 100001, (1 << 20) + 7]:
        for off in (0, 1, 3):
            big = rng.gamma(30, 1.0, n + 17) * rng.choice([1e-3, 1.0, 1e3], n + 17)
            v = big[off:off + n]
            g = np.where(rng.random(n) < 0.2, -1, 40).astype(np.int8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want, wantm = np.average(v), np.average(v[g > -1])
            got = oracle.np_mean(v)
            gotm, k = oracle.np_masked_mean(v, g)
            assert got == want or (np.isnan(got) and np.isnan(want)), (n, off)
            assert (gotm == wantm or (np.isnan(gotm) and np.isnan(wantm))) and k == int((g > -1).sum()), (n, off)


def test_y_pass_on_arbitrary_labels_pinned_to_the_reference(golden_dir):
    """oracle.y_coordinate_clustering_labels == the real DBSCAN.y_coordinate_clustering on labels of every shape (another eps/m,
    clusters above 128 members, a value on several ranges, sparse / float / negative values, cluster_id below the largest label)"""
    import oracle
    z = np.load(os.path.join(golden_dir, "dbscan_y_labels.npz"))
    n = sum(1 for k in z.files if k.endswith("_par"))
    kinds = set()
    for c in range(n):
        eps, m, cid, want_id = (int(v) for v in z["c%d_par" % c])
        got, gid = oracle.y_coordinate_clustering_labels(z["c%d_data" % c], eps, m, cid, z["c%d_in" % c])
        assert gid == want_id and np.array_equal(got, z["c%d_out" % c]), (c, str(z["c%d_kind" % c]))
        kinds.add(str(z["c%d_kind" % c]))
    assert kinds == {"other_eps", "large", "split_ranges", "arbitrary", "collide", "collide_random", "many_sparse"}
    # the collision cases really differ from the collision-free closed form (else they would test nothing)
    differ = 0
    for c in range(n):
        if str(z["c%d_kind" % c]).startswith("collide"):
            eps, m, cid, _ = (int(v) for v in z["c%d_par" % c])
            lab = z["c%d_in" % c]
            hi = max(cid, int(lab.max())) + 1000
            shifted, _ = oracle.y_coordinate_clustering_labels(z["c%d_data" % c], eps, m, hi, lab)
            a, b = z["c%d_out" % c], shifted
            same_partition = len(set(zip(a.tolist(), b.tolist()))) == len(set(a.tolist())) == len(set(b.tolist()))
            differ += not same_partition
    assert differ >= 3
