"""Multi-GPU sharding of the clustering path.

(chrA,chrB) signal buckets are independent (tiddit_cluster.pyx:140-154 keeps no cross-bucket state),
so they are bin-packed onto ranks (one process per GPU) and clustered without any data-path
collective; ONE exchange step — a variable-count all-gather of the label arrays over RCCL/xGMI —
assembles the final cluster set on every rank.  The coverage / GC histograms shard by contig and
need no exchange (each rank returns its contigs' bins to the host caller).
"""


def shard_buckets(sizes, world_size):
    """Longest-processing-time bin packing of buckets by signal count.
    -> list (per rank) of bucket indices, each ascending; deterministic on every rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    owned = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += int(sizes[i])
    return [sorted(o) for o in owned]


def shard_contigs(lengths, world_size):
    """Same packing for the histogram path (work ~ contig length)."""
    return shard_buckets(lengths, world_size)


def allgatherv(t, group=None):
    """Variable-count all-gather of 1-D tensors (RCCL has no allgatherv): counts are exchanged
    first, payloads are padded to the maximum and gathered with ONE all_gather_into_tensor.
    -> list of world_size tensors (rank order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cnt = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    counts = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.cpu().tolist()
    mx = max(max(counts), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = torch.empty(world * mx, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx:r * mx + counts[r]] for r in range(world)]


def split_gathered(bucket_sizes, owned, parts):
    """per-rank label arrays (rank r: its buckets in owned[r] order, concatenated) -> list over ALL buckets"""
    labels = [None] * len(bucket_sizes)
    for r, p in enumerate(parts):
        off = 0
        for b in owned[r]:
            labels[b] = p[off:off + int(bucket_sizes[b])]
            off += int(bucket_sizes[b])
    return labels


def cluster_buckets_distributed(bucket_sizes, cluster_local, group=None, wire_dtype="int32", flat=False):
    """Cluster every bucket on its owner rank, then all-gather the labels.

    Labels are integers (-1 or a cluster id < 2^31) carried in float64 like the reference returns them; on
    the wire they travel as int32 (half the xGMI bytes, exact) unless wire_dtype is None.

    bucket_sizes : number of signals of every bucket (identical on all ranks)
    cluster_local(bucket_ids) -> 1-D float64 tensor with the labels of those buckets, concatenated
                   in the given order (on the GPU path: tdt_dbscan_device over the rank's buckets)
    -> list over ALL buckets of label tensors, identical on every rank (flat=True: (owned, per-rank wire arrays) instead).
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owned = shard_buckets(bucket_sizes, world)
    import torch
    mine = cluster_local(owned[rank])
    out_dtype = mine.dtype
    if wire_dtype is not None:
        mine = mine.to(getattr(torch, wire_dtype))
    parts = allgatherv(mine, group)
    if flat:                       # the gathered per-rank arrays as they arrived (split_gathered gives the per-bucket view)
        return owned, parts
    return split_gathered(bucket_sizes, owned, [p.to(out_dtype) for p in parts])


# ------------------------------------------------------------------------------------------------------------------
# One BAM, N GPUs: every rank ingests the BGZF blocks of its byte range of the file (bamio.DeviceBamReader(shard=...)),
# accumulates its reads into a full-genome histogram, and ONE all-reduce of the float64 bins assembles the result.
# The bins are exact multiples of 2^-S far below 2^53, so the sum is exact and order independent:
# the reduced array is bit-identical to the single-GPU one.

def check_seams(first_off, next_off, empty, group=None):
    """Exactness of a sharded read.  Shard 0 starts at the header (known offset); shard r > 0 guessed its first record
    `first_off` bytes into its range, and shard r-1 — following the true block_size chain across the seam — reports
    where that record must start (`next_off`).  Agreement at every seam (empty shards pass the value through) makes
    every shard's decode the sequential one.  Raises ValueError on any disagreement; returns the gathered table."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([-1 if first_off is None else int(first_off), -1 if next_off is None else int(next_off), 1 if empty else 0],
                        dtype=torch.int64, device=dev)
    world = dist.get_world_size(group)
    table = torch.empty(3 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table, mine, group=group)
    table = table.cpu().view(world, 3).tolist()
    expect = None
    for r, (fo, no, emp) in enumerate(table):
        if emp:
            continue
        if expect is not None and fo != expect:
            raise ValueError("sharded BAM read: shard %d starts its records at +%d, the chain of shard before it says +%d" % (r, fo, expect))
        expect = no
    return table


def allreduce_bins(bins, group=None):
    """In-place SUM all-reduce of a float64 tensor of bins (RCCL over xGMI with the nccl backend)."""
    import torch.distributed as dist
    dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)
    return bins


def coverage_sharded(bam_file_name, bin_size, min_q, group=None, ctx=None, chunk=448 << 20):
    """`tiddit --cov` on one BAM with one process per GPU.  -> (header, {contig: float64 bins}) on every rank."""
    import torch
    import torch.distributed as dist
    from . import _native, tiddit_coverage
    from .bamio import DeviceBamReader
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx = ctx or _native.default_context()
    reader = DeviceBamReader(bam_file_name, ctx=ctx, chunk=chunk, shard=(rank, world))
    header = reader.header
    hist = tiddit_coverage.CoverageHistogram(header, bin_size, ctx=ctx)
    reader.bin_for(hist)
    n = 0
    for b in reader.batches():
        hist.push_device_batch(b, min_q)
        reader.ahead()                           # the next span's inflate follows the launch (which reads coverage records, not raw bytes)
        n += len(b)
    empty = reader.first_off is None
    first_off, next_off = reader.first_off, reader.next_off
    reader.close()
    check_seams(first_off, next_off, empty, group)
    dev = torch.device("cuda", ctx.device)
    bins = torch.empty(hist.total_bins(), dtype=torch.float64, device=dev)   # every bin is written by the finalize kernel
    torch.cuda.synchronize(dev)          # torch's allocator/fill work runs on torch's stream, the library on its own: order them
    hist.finish_all_device(bins.data_ptr())
    ctx.sync()
    if dist.get_backend(group) != "nccl":
        bins = allreduce_bins(bins.cpu(), group)
    else:
        allreduce_bins(bins, group)
    host = bins.cpu().numpy()
    out = {}
    for i, c in enumerate(header["SQ"]):
        o = hist.offset(i)
        out[c["SN"]] = host[o:o + hist.nbins(i)[0]].copy()
    hist.close()
    return header, out, n


# ------------------------------------------------------------------------------------------------------------------
# `tiddit --sv` on N ranks (BASELINE configs[4]): small host objects (library statistics, signal rows) travel as byte tensors
# through the process group, so the same code runs over RCCL (device tensors) and gloo (host tensors).

def _wire_device(group=None):
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _wire_max():
    """largest single message of the byte exchanges below (TIDDIT_WIRE_MAX_BYTES, default 1 GiB): longer payloads travel as several —
    RCCL / gloo counts beyond 2^31 elements are where transports have failed before, and a 54-GB BAM's rows on few ranks get there"""
    import os
    return max(1, int(os.environ.get("TIDDIT_WIRE_MAX_BYTES", str(1 << 30))))


def broadcast_object(obj, src=0, group=None):
    """pickle -> uint8 tensor -> broadcast (length first).  -> the object on every rank."""
    import pickle
    import numpy
    import torch
    import torch.distributed as dist
    dev = _wire_device(group)
    me = dist.get_rank(group)
    blob = pickle.dumps(obj, protocol=4) if me == src else b""
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src, group=group)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if me == src and len(blob):
        buf.copy_(torch.from_numpy(numpy.frombuffer(blob, dtype=numpy.uint8).copy()))
    cap = _wire_max()
    for o in range(0, buf.numel(), cap):
        dist.broadcast(buf[o:o + cap], src, group=group)
    return obj if me == src else pickle.loads(buf.cpu().numpy().tobytes())


def gather_bytes(blob, dst=0, group=None):
    """variable-length byte strings of every rank -> list in RANK ORDER on `dst` (None elsewhere): lengths by one all-gather,
    payloads point to point (no padding to the largest rank)."""
    import numpy
    import torch
    import torch.distributed as dist
    dev = _wire_device(group)
    me, world = dist.get_rank(group), dist.get_world_size(group)
    cnt = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.cpu().tolist()
    cap = _wire_max()
    if me != dst:
        if len(blob):
            t = torch.from_numpy(numpy.frombuffer(blob, dtype=numpy.uint8).copy()).to(dev)
            for o in range(0, t.numel(), cap):
                dist.send(t[o:o + cap], dst, group=group)
        return None
    out = []
    for r in range(world):
        if r == me:
            out.append(bytes(blob))
            continue
        buf = torch.empty(counts[r], dtype=torch.uint8, device=dev)
        for o in range(0, counts[r], cap):
            dist.recv(buf[o:o + cap], r, group=group)
        out.append(buf.cpu().numpy().tobytes())
    return out


def contig_owners(lengths, kept, world_size):
    """owner rank of every contig id for the N-rank signal tables: the contigs >= min_contig (`kept`) bin-packed by length (the rows
    of a pair (chrA, chrB) are almost all intra-chromosomal, so a chrA's share of the work follows its length); every (chrA, *) pair
    lives on owner[chrA].  -> int32 array, identical on every rank; contigs that own nothing get rank 0."""
    import numpy
    ids = [i for i, k in enumerate(kept) if k]
    owner = numpy.zeros(len(lengths), dtype=numpy.int32)
    for r, mine in enumerate(shard_contigs([lengths[i] for i in ids], world_size)):
        for j in mine:
            owner[ids[j]] = r
    return owner


def allgather_i64(values, group=None):
    """int64 vector of every rank -> [world, len] numpy array (rank order)"""
    import numpy
    import torch
    import torch.distributed as dist
    dev = _wire_device(group)
    world = dist.get_world_size(group)
    mine = torch.from_numpy(numpy.ascontiguousarray(values, dtype=numpy.int64)).to(dev)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().reshape(world, -1)


def alltoall_bytes(parts, group=None):
    """parts[r] = the uint8 array this rank has for rank r -> the arrays every rank had for THIS rank, in RANK ORDER.  Sizes by one
    all-gather; payloads by ONE all_to_all_single over RCCL (device tensors), or point to point over gloo (which has no all-to-all):
    every receive is posted before the sends."""
    import numpy
    import torch
    import torch.distributed as dist
    me, world = dist.get_rank(group), dist.get_world_size(group)
    parts = [numpy.ascontiguousarray(p, dtype=numpy.uint8) for p in parts]
    sizes = allgather_i64([len(p) for p in parts], group)            # sizes[src][dst]
    incoming = [int(sizes[r][me]) for r in range(world)]
    cap = _wire_max()
    rounds = -(-int(sizes.max()) // cap) if sizes.size and sizes.max() > 0 else 0      # (every rank holds the whole size table: same count everywhere)
    if dist.get_backend(group) == "nccl":
        dev = _wire_device(group)
        got = [numpy.empty(n, dtype=numpy.uint8) for n in incoming]
        for k in range(rounds):                                  # one all_to_all_single per `cap` bytes of the largest (source, destination) piece
            lo = k * cap
            out_sizes = [len(p[lo:lo + cap]) for p in parts]
            in_sizes = [max(0, min(cap, n - lo)) for n in incoming]
            send = torch.from_numpy(numpy.concatenate([p[lo:lo + cap] for p in parts]) if sum(out_sizes) else numpy.zeros(0, dtype=numpy.uint8)).to(dev)
            recv = torch.empty(sum(in_sizes), dtype=torch.uint8, device=dev)
            dist.all_to_all_single(recv, send, output_split_sizes=in_sizes, input_split_sizes=out_sizes, group=group)
            flat = recv.cpu().numpy()
            o = 0
            for r, n in enumerate(in_sizes):
                got[r][lo:lo + n] = flat[o:o + n]
                o += n
        return got
    bufs, reqs = [], []
    for r in range(world):
        if r == me:
            bufs.append(parts[me])
            continue
        buf = torch.empty(incoming[r], dtype=torch.uint8)
        for o in range(0, incoming[r], cap):
            reqs.append(dist.irecv(buf[o:o + cap], r, group=group))
        bufs.append(buf.numpy())
    for r in range(world):
        if r != me:
            t = torch.from_numpy(parts[r])
            for o in range(0, len(parts[r]), cap):
                reqs.append(dist.isend(t[o:o + cap], r, group=group))
    for q in reqs:
        q.wait()
    return bufs


# One oversized (chrA,chrB) bucket cut into pieces that cluster independently.  A cut is legal between two
# posA-neighbours a < b with b - a >= eps: every sliding window that spans the gap fails `max(distances) < epsilon`
# (DBSCAN.py:50), so no x-run — hence no cluster — crosses it.  ONE detail makes a naive cut wrong: the reference's last
# window of an array is one point short (`data[i+1:i+m+1]` at i = n-m, DBSCAN.py:41-43), so the END of an array is treated more
# leniently than the same position in the middle of one.  Every piece but the last therefore carries one HALO point — a
# coordinate >= eps beyond its last signal, standing in for the first signal of the next piece — which makes its final
# windows exactly the windows of the uncut array; the halo itself can never be labelled (it is >= m+1 positions behind the last
# passing window start) and is dropped.  Ids are then re-based by exclusive scans of the pieces' x-run and extra-sub-run
# counts (DBSCAN.py:112-122: extra sub-runs are numbered after ALL x-runs of the array).

def plan_bucket_cuts(posA, eps, parts, max_cells=1 << 24):
    """-> (thresholds, halo_width): ascending cut values t (piece k holds t[k-1] <= posA < t[k]) such that no signal lies in
    [t, t + halo_width) and halo_width >= ceil(eps); [] when the bucket cannot (or need not) be cut.  Host planner, O(n):
    an occupancy histogram over cells of >= ceil(eps) bp — an empty cell is a legal gap — and the empty cell nearest each
    1/parts quantile.  Deterministic: every rank computes the same plan."""
    import math
    import numpy
    posA = numpy.asarray(posA, dtype=numpy.int64)
    n = len(posA)
    if not (eps > 0) or parts < 2 or n < 2 * parts:
        return [], 0
    E = int(math.ceil(eps))
    lo, hi = int(posA.min()), int(posA.max())
    W = max(E, -(-(hi - lo + 1) // max_cells))
    cells = (posA - lo) // W
    cnt = numpy.bincount(cells, minlength=(hi - lo) // W + 1)
    empty = numpy.flatnonzero(cnt == 0)
    if len(empty) == 0:
        return [], 0
    before = numpy.cumsum(cnt)[empty]                    # signals below the empty cell (non-decreasing along `empty`)
    chosen = set()
    for k in range(1, parts):
        target = n * k / parts
        j = int(numpy.searchsorted(before, target))
        best = min((c for c in (j - 1, j) if 0 <= c < len(empty)), key=lambda c: (abs(before[c] - target), c))
        if 0 < before[best] < n:
            chosen.add(int(before[best]))                # one cut per distinct split of the signals
    ts = []
    for b in sorted(chosen):
        c = empty[int(numpy.searchsorted(before, b))]    # the first empty cell with that many signals below it
        ts.append(lo + int(c) * W)
    return ts, W
