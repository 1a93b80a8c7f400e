"""Property tests (CPU, hypothesis): the two oracle formulations agree, the closed form the kernels implement
(restated here in numpy) equals the literal state machine, and the C BAM decoder
agrees with the independent pure-Python parser on random records."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle


def closed_form_x(x, eps, m):
    """labels of x_coordinate_clustering for SORTED x via the closed form: p[i] = x[min(i+m,n-1)] - x[i] < eps for
    i <= n-m, runs of true p numbered 0.., label[k] = run of the last true p within m-1 positions back."""
    n = len(x)
    lab = np.full(n, -1.0)
    if n < m:
        return lab, -1
    p = np.zeros(n, dtype=bool)
    for i in range(n - m + 1):
        p[i] = x[min(i + m, n - 1)] - x[i] < eps
    start = p & ~np.concatenate([[False], p[:-1]])
    run = np.cumsum(start) - 1
    for k in range(n):
        for j in range(k, max(-1, k - m), -1):
            if p[j]:
                lab[k] = run[j]
                break
    return lab, int(run[-1]) if start.any() else -1


@settings(max_examples=150, deadline=None)
@given(st.lists(st.integers(0, 400), min_size=0, max_size=60), st.integers(1, 60), st.integers(2, 6))
def test_closed_form_equals_state_machine(xs, eps, m):
    x = np.sort(np.array(xs, dtype=np.int64))
    data = np.stack([x, x], 1).reshape(len(x), 2)
    want, wid = oracle.x_coordinate_clustering(data, eps, m)
    got, gid = closed_form_x(x, eps, m)
    assert np.array_equal(got, want) and gid == wid


@settings(max_examples=150, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 300), st.integers(0, 300)), min_size=1, max_size=80), st.integers(1, 80), st.integers(2, 5))
def test_literal_and_sweep_y_pass_agree(pts, eps, m):
    a = np.array(sorted(pts, key=lambda t: t[0]), dtype=np.int64).reshape(len(pts), 2)
    lit = oracle.dbscan_main(a, eps, m, literal=True)
    swp = oracle.dbscan_main(a, eps, m, literal=False)
    assert np.array_equal(lit, swp)
    # labels are either -1 or dense non-negative ids, and every cluster id that exists has >= 1 member
    ids = np.unique(lit[lit >= 0])
    assert np.all(ids == np.floor(ids))


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 5000), st.integers(1, 700)), min_size=0, max_size=200), st.sampled_from([1, 7, 50, 500, 4096]))
def test_coverage_order_independent(reads, z):
    """the float64 sum of the reference is exact (float32 addends), hence independent of the read order"""
    LN = 6000
    s = np.array([r[0] for r in reads], dtype=np.int64)
    e = np.minimum(s + np.array([r[1] for r in reads], dtype=np.int64), LN)
    mq = np.full(len(s), 60, np.uint8)
    fl = np.zeros(len(s), np.uint16)
    a, _ = oracle.coverage_stream(s, e, mq, fl, LN, z, 0)
    perm = np.random.default_rng(len(reads)).permutation(len(s))
    b, _ = oracle.coverage_stream(s[perm], e[perm], mq, fl, LN, z, 0)
    assert np.array_equal(a, b)


cigar_op = st.tuples(st.sampled_from("MIDNSH=X"), st.integers(1, 300))


@settings(max_examples=25, deadline=None)
@given(recs=st.lists(st.tuples(st.integers(0, 90000), st.integers(0, 60), st.integers(0, 0xfff), st.lists(cigar_op, min_size=0, max_size=5),
                               st.integers(-1, 1), st.integers(-100000, 100000), st.booleans()), min_size=0, max_size=40))
def test_bam_decoder_random_records(tmp_path_factory, recs):
    from oracle import signal_oracle
    from tiddit_amd import bamio, build
    build.build()
    path = str(tmp_path_factory.mktemp("fz") / "f.bam")
    w = bamio.BamWriter(path, [("c1", 100000), ("c2", 5000)])
    for i, (pos, mapq, flag, cig, mtid, tlen, sa) in enumerate(sorted(recs, key=lambda r: r[0])):
        cigar = "".join("%d%s" % (l, op) for op, l in cig)
        tags = [("SA", "Z", "c2,%d,+,30M,60,0;" % (i + 1))] if sa else []
        w.write("q%d" % i, flag, 0, pos, mapq, cigar, mtid, pos, tlen, seq="ACGTN"[:i % 6], tags=tags)
    w.close()
    hdr, reads = signal_oracle.parse_bam(path)
    got = []
    for b in bamio.BamReader(path).batches():
        for i in range(len(b)):
            got.append((int(b.pos[i]), int(b.end[i]), int(b.mapq[i]), int(b.flag[i]), int(b.mate_tid[i]), int(b.tlen[i]),
                        int(b.sa_off[i]) >= 0, b.record(i).query_name))
    want = [(r.reference_start, r.reference_end, r.mapq, r.flag, r.next_reference_id, r.isize, "SA" in r.tags, r.query_name) for r in reads]
    assert got == want
