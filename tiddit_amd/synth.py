"""Deterministic synthetic inputs for the hot path.

These are *input specifications*, not reference code: a coordinate-sorted 30x-WGS-shaped
alignment stream for the coverage histogram, a planted-cluster (posA,posB) point cloud for the
clustering path, and a random reference sequence for the GC histogram.  The golden checksums in
``tests/golden`` were produced from exactly these generators (seed 20260928).
"""
import numpy as np

SEED = 20260928


def gen_reads(L, depth, read_len=150, seed=SEED):
    """One contig's coordinate-sorted read stream -> (start i64, end i64, mapq u8, flag u16).

    ``end`` is the 0-based exclusive reference end (htslib ``bam_endpos``).  5 % of the reads are
    soft-clipped (shorter reference span), 5 % carry a deletion (longer span); flags mix
    duplicate / unmapped / supplementary / secondary bits so the a3/a4 read filter is exercised.
    """
    rng = np.random.default_rng(seed)
    n = int(L * depth / read_len)
    start = np.sort(rng.integers(0, L - read_len + 1, n)).astype(np.int64)
    kind = rng.random(n)
    reflen = np.full(n, read_len, dtype=np.int64)
    clip = rng.integers(1, 60, n)
    dele = rng.integers(1, 30, n)
    reflen = np.where(kind < 0.05, read_len - clip, reflen)
    reflen = np.where((kind >= 0.05) & (kind < 0.10), read_len + dele, reflen)
    end = np.minimum(start + reflen, L).astype(np.int64)
    mapq = rng.choice(np.array([0, 1, 10, 20, 30, 60], dtype=np.uint8), size=n,
                      p=[.04, .01, .02, .03, .10, .80])
    u = rng.random((5, n))
    flag = ((0x1 | 0x2) | np.where(u[0] < .5, 0x10, 0x20) | np.where(u[1] < .02, 0x400, 0)
            | np.where(u[2] < .005, 0x4, 0) | np.where(u[3] < .01, 0x800, 0)
            | np.where(u[4] < .005, 0x100, 0))
    return start, end, mapq, flag.astype(np.uint16)


def gen_points(n, L=2_000_000_000, seed=SEED, frac_clustered=0.6, per_cluster=6, jitter=300):
    """(posA,posB,signal_index) int64 [n,3] for ONE chr pair, stably sorted by posA — the order
    ``tiddit_cluster.pyx:152`` hands to ``DBSCAN.main``."""
    rng = np.random.default_rng(seed)
    k = int(n * frac_clustered) // per_cluster
    cx = rng.integers(1000, L - 200_000, k)
    cy = cx + rng.integers(1000, 100_000, k)
    nc = k * per_cluster
    A = np.concatenate([np.repeat(cx, per_cluster) + rng.integers(0, jitter, nc),
                        rng.integers(1000, L - 1000, n - nc)])
    B = np.concatenate([np.repeat(cy, per_cluster) + rng.integers(0, jitter, nc),
                        rng.integers(1000, L - 1000, n - nc)])
    perm = rng.permutation(n)
    A = A[perm]
    B = B[perm]
    pts = np.stack([A, B, np.arange(n)], 1).astype(np.int64)
    return pts[np.argsort(pts[:, 0], kind="stable")]


def gen_sequence(L, seed=SEED, n_frac=0.02, lower_frac=0.3, gc=0.41):
    """Random reference-like ASCII sequence (uint8): ACGT with the given GC content, runs of N
    (hard-masked gaps), soft-masked lowercase stretches and a sprinkle of IUPAC codes."""
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=L, p=p)].copy()
    # IUPAC sprinkle
    k = max(1, L // 5000)
    seq[rng.integers(0, L, k)] = np.frombuffer(b"RYSWKM", dtype=np.uint8)[rng.integers(0, 6, k)]
    # N runs
    n_runs = max(1, int(L * n_frac) // 400)
    for s, ln in zip(rng.integers(0, L, n_runs), rng.integers(1, 800, n_runs)):
        seq[s:s + ln] = ord("N")
    # lowercase stretches
    l_runs = max(1, int(L * lower_frac) // 300)
    for s, ln in zip(rng.integers(0, L, l_runs), rng.integers(1, 600, l_runs)):
        seq[s:s + ln] |= 0x20
    return seq


def gen_reads_device(L, depth, device, read_len=150, seed=SEED):
    """Same distribution as :func:`gen_reads`, generated directly in HBM with torch (different RNG
    stream, so not the golden-checksum input): -> (start i32, end i32, mapq u8, flag i16-as-u16 bits)
    tensors on ``device``.  Used by bench.py for the 600 M-read config-2 stream."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = int(L * depth / read_len)
    start = torch.randint(0, L - read_len + 1, (n,), generator=g, device=device, dtype=torch.int32)
    start, _ = torch.sort(start)
    kind = torch.rand(n, generator=g, device=device)
    clip = torch.randint(1, 60, (n,), generator=g, device=device, dtype=torch.int32)
    dele = torch.randint(1, 30, (n,), generator=g, device=device, dtype=torch.int32)
    reflen = torch.full((n,), read_len, device=device, dtype=torch.int32)
    reflen = torch.where(kind < 0.05, read_len - clip, reflen)
    reflen = torch.where((kind >= 0.05) & (kind < 0.10), read_len + dele, reflen)
    end = torch.minimum(start + reflen, torch.tensor(L, device=device, dtype=torch.int32))
    del kind, clip, dele, reflen
    u = torch.rand(n, generator=g, device=device)
    cdf = torch.tensor([.04, .05, .07, .10, .20], device=device)
    vals = torch.tensor([0, 1, 10, 20, 30, 60], device=device, dtype=torch.uint8)
    mapq = vals[torch.bucketize(u, cdf, right=True)]
    flag = torch.full((n,), 0x3, device=device, dtype=torch.int16)
    u = torch.rand(n, generator=g, device=device)
    flag |= torch.where(u < .5, 0x10, 0x20).to(torch.int16)
    for p, bit in ((.02, 0x400), (.005, 0x4), (.01, 0x800), (.005, 0x100)):
        u = torch.rand(n, generator=g, device=device)
        flag |= torch.where(u < p, bit, 0).to(torch.int16)
    return start, end, mapq, flag
