"""Drop-in for ``tiddit.tiddit_cluster`` (tiddit_cluster.pyx) on the MI355X.

``main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig,
skip_assembly, min_reads) -> candidates`` (:39-338) and ``find_discordant_pos(fragment, is_mp)`` (:7-37)
keep the reference's arguments and return the same nested ``candidates[chrA][chrB][cluster_id]``
dictionaries (same keys, same insertion order — the order later defines the VCF ``SV_n`` ids).

What moved to the GPU: the stable sort of every (chrA,chrB) bucket by posA and ``DBSCAN.main`` on it
(:152-154) — all buckets in ONE ``tdt_cluster_columns`` call instead of a serial Python loop.  The signal table
(:47-137) and the per-signal half of the regrouping (:156-254) come from the native signal tables when
``tiddit_signal.main`` of this process wrote the files (``sigtab.SignalTables``: columns written into pinned
buffers, candidates' members returned as flat arrays — Python touches candidates, not rows); files from
elsewhere are parsed as text by the literal loops below.  Reference quirks reproduced on purpose are marked QUIRK.
"""
from collections import Counter  # noqa: F401  (kept: callers of the reference module may import it from here)

import os

import numpy

from .hostutil import quiet_gc
from . import _native

try:                                  # the CPython extension that builds the candidate dictionaries (csrc/tdt_pycand.c, built by tiddit_amd.build);
    from . import _pycand             # without it — no interpreter headers where the package was built — the Python loop below does the same
except ImportError:
    _pycand = None

# (orientation of read A, orientation of read B) -> which of (start, end) is the breakpoint side,
# indexes into the 9-column discordants_*.tab row: 3=startA 4=endA 6=startB 7=endB   (:7-37)
_PE_SIDE = {("False", "True"): (4, 6), ("False", "False"): (4, 7), ("True", "True"): (3, 6)}
_MP_SIDE = {("False", "True"): (3, 7), ("False", "False"): (3, 6), ("True", "True"): (4, 7)}


def find_discordant_pos(fragment, is_mp):
    """Pick (posA, posB) of a discordant pair from its read coordinates and orientations."""
    key = (fragment[5], fragment[8])
    if is_mp:
        a, b = _MP_SIDE.get(key, (4, 6))
    else:
        a, b = _PE_SIDE.get(key, (3, 7))
    return (fragment[a], fragment[b])


def _new_candidate():
    side = lambda: {"contigs": [], "splits": [], "discordants": [], "orientation_contigs": [], "orientation_splits": [],
                    "orientation_discordants": [], "start": [], "end": []}
    c = {"signal_type": {}, "samples": set([]), "sample_discordants": {}, "sample_splits": {}, "sample_contigs": {},
         "N_discordants": 0, "discordants": set([]), "N_splits": 0, "splits": set([]), "N_contigs": 0, "contigs": set([]),
         "n_signals": 0, "posA": 0}
    c["positions_A"] = side()
    c["start_A"] = 0
    c["end_A"] = 0
    c["posB"] = 0
    c["positions_B"] = side()
    c["start_B"] = 0
    c["end_B"] = 0
    return c


_KIND = {"D": "discordants", "S": "splits", "A": "contigs"}


STAGE_SECONDS = {}          # wall seconds of the last main(), stage by stage


def _read_signals(prefix, samples, contig_length, is_mp, min_contig, skip_assembly):
    """Parse discordants_/splits_/contigs_{sample}.tab in the reference's order (:46-137) — the text way in (files this process did
    not write, or --with assembly contigs); the job's own tables never become text rows (:func:`_main_native`).
    -> signals[chrA][chrB] = list of records, positions[chrA][chrB] = flat list posA, posB, i, posA, posB, i, ... (one array
    conversion per bucket later, not one per row)."""
    signals, positions = {}, {}
    pos_of = {}                  # id(record list of a bucket) -> its flat position list
    i = 0

    def bucket(chrA, chrB):
        recs = signals.setdefault(chrA, {}).get(chrB)
        if recs is None:
            recs = signals[chrA][chrB] = []
            pos_of[id(recs)] = positions.setdefault(chrA, {}).setdefault(chrB, [])
        return recs

    for sample in samples:
        disc_path, split_path = "{}_tiddit/discordants_{}.tab".format(prefix, sample), "{}_tiddit/splits_{}.tab".format(prefix, sample)
        disc_iter = (line.rstrip().split("\t") for line in open(disc_path))
        for c in disc_iter:
            chrA, chrB = c[1], c[2]
            if contig_length[chrA] < min_contig or contig_length[chrB] < min_contig:
                continue
            recs = bucket(chrA, chrB)
            posA, posB = find_discordant_pos(c, is_mp)
            if int(posA) > contig_length[chrA]:
                posA = contig_length[chrA]
                if int(posB) > contig_length[chrB]:
                    posA = contig_length[chrB]      # QUIRK (:67-70): posB is never clipped, posA takes chrB's length
            recs.append([c[0], sample, "D", posA, c[5], posB, c[8], i, int(c[3]), int(c[4]), int(c[6]), int(c[7])])
            pos_of[id(recs)].extend((int(posA), int(posB), i))
            i += 1
        files = [("S", "{}_tiddit/splits_{}.tab")]
        if not skip_assembly:
            files.append(("A", "{}_tiddit/contigs_{}.tab"))
        for kind, pattern in files:
            rows_iter = (line.rstrip().split("\t") for line in open(pattern.format(prefix, sample)))
            for c in rows_iter:
                chrA, chrB = c[1], c[2]
                if contig_length[chrA] < min_contig or contig_length[chrB] < min_contig:
                    continue
                recs = bucket(chrA, chrB)
                posA, posB = c[3], c[5]
                if int(posA) > contig_length[chrA]:
                    posA = contig_length[chrA]
                if int(posB) > contig_length[chrB]:
                    posB = contig_length[chrB]
                recs.append([c[0], sample, kind, posA, c[4], posB, c[6], i, int(c[7]), int(c[8]), int(c[9]), int(c[10])])
                pos_of[id(recs)].extend((int(posA), int(posB), i))
                i += 1
    return signals, positions


_POOL = None


def cluster_buckets(buckets, epsilon, m, ctx=None, counts=False, max_pos=0):
    """buckets: list of integer [n_b, >=2] arrays (posA, posB, ...) in signal order.
    -> list of float64 label arrays, labels[b][j] = cluster of the bucket's j-th signal
    (= DBSCAN.main on the bucket stably sorted by posA, mapped back; tiddit_cluster.pyx:152-160).
    counts=True: -> (labels, x-runs per bucket, final cluster_id per bucket).
    The columns go to the device as int32 built in pinned memory (``tdt_cluster_columns``: no host pass over them, labels come
    back in signal order, 4 B each); max_pos (e.g. the longest contig) bounds posA so that only its significant digits are
    sorted.  Coordinates outside int32 take the int64 entry point (``tdt_sort_dbscan_ex``)."""
    global _POOL
    ctx = ctx or _native.default_context()
    sizes = [len(b) for b in buckets]
    n = int(sum(sizes))
    off = numpy.zeros(len(buckets) + 1, dtype=numpy.int64)
    numpy.cumsum(sizes, out=off[1:])
    runs = numpy.zeros(len(buckets), dtype=numpy.int64)
    last = numpy.full(len(buckets), -1, dtype=numpy.int64)
    if n == 0:
        labs = [numpy.zeros(0) for _ in buckets]
        return (labs, runs, last) if counts else labs
    arrs = [numpy.asarray(b) for b in buckets]
    for a in arrs:
        if len(a) and not numpy.issubdtype(a.dtype, numpy.integer):
            raise TypeError("signal positions must be integers (got dtype %s)" % a.dtype)
    lo = min(int(a[:, :2].min()) for a in arrs if len(a))
    hi = max(int(a[:, :2].max()) for a in arrs if len(a))
    if lo >= -(1 << 31) and hi < (1 << 31):
        if _POOL is None:
            from .hostutil import PinnedPool
            _POOL = PinnedPool()
        posA, posB, lab32 = (_POOL.take(k, n, numpy.int32) for k in ("posA", "posB", "labels"))
        for a, o in zip(arrs, off[:-1]):
            if len(a):
                posA[o:o + len(a)] = a[:, 0]
                posB[o:o + len(a)] = a[:, 1]
        bound = int(max_pos) if max_pos and lo >= 0 and hi <= max_pos else (hi if lo >= 0 else 0)
        _native.check(ctx.lib.tdt_cluster_columns(ctx.handle, _native.ptr(posA), _native.ptr(posB), n, _native.ptr(off), len(buckets), float(epsilon),
                                                  int(m), bound, _native.ptr(lab32), _native.ptr(runs) if counts else None,
                                                  _native.ptr(last) if counts else None))
        by_signal = lab32.astype(numpy.float64)
    else:
        posA = numpy.ascontiguousarray(numpy.concatenate([a[:, 0] for a in arrs if len(a)]), dtype=numpy.int64)
        posB = numpy.ascontiguousarray(numpy.concatenate([a[:, 1] for a in arrs if len(a)]), dtype=numpy.int64)
        perm = numpy.empty(n, dtype=numpy.uint32)
        lab = numpy.empty(n, dtype=numpy.float64)
        _native.check(ctx.lib.tdt_sort_dbscan_ex(ctx.handle, _native.ptr(posA), _native.ptr(posB), n, _native.ptr(off), len(buckets),
                                                 float(epsilon), int(m), _native.ptr(perm), _native.ptr(lab), _native.ptr(runs), _native.ptr(last)))
        by_signal = numpy.empty(n, dtype=numpy.float64)
        by_signal[perm] = lab
    labs = [by_signal[off[b]:off[b + 1]] for b in range(len(buckets))]
    return (labs, runs, last) if counts else labs


def plan_pieces(buckets, epsilon, world, balance=1.0, min_cut=32768):
    """The work list of the N-rank clustering step: every bucket, or — for a bucket above `balance` x (all signals / world) —
    (and at least `min_cut` signals: below that a launch is latency, not work) its pieces cut at posA gaps >= epsilon
    (dist.plan_bucket_cuts).  -> list of (bucket index, piece number, member indices
    or None for a whole bucket, halo posA or None), in (bucket, piece) order; identical on every rank."""
    from .dist import plan_bucket_cuts
    sizes = [len(b) for b in buckets]
    total = sum(sizes)
    pieces = []
    for b, bucket in enumerate(buckets):
        ts = []
        if world > 1 and total and sizes[b] >= min_cut and sizes[b] > balance * total / world:
            posA = numpy.asarray(bucket, dtype=numpy.int64)[:, 0]
            parts = min(world * 2, max(2, int(round(sizes[b] * world / total))) * 2)      # pieces of about half a rank's fair share
            ts, width = plan_bucket_cuts(posA, epsilon, parts)
        if not ts:
            pieces.append((b, 0, None, None))
            continue
        which = numpy.searchsorted(numpy.asarray(ts, dtype=numpy.int64), posA, side="right")
        for k in range(len(ts) + 1):
            pieces.append((b, k, numpy.flatnonzero(which == k), (ts[k] + width) if k < len(ts) else None))
    return pieces


def rebase_pieces(piece_labels, piece_runs, piece_last):
    """ids of the pieces of ONE bucket (each numbered from 0 by its own DBSCAN.main) -> the ids DBSCAN.main gives the uncut bucket
    (DBSCAN.py:112-122): x-run r of piece k becomes r + (x-runs of the pieces before it); an extra sub-run — local id >= the piece's
    x-run count — is numbered after ALL x-runs of the bucket, behind the extra sub-runs of the pieces before it."""
    runs = numpy.asarray(piece_runs, dtype=numpy.int64)
    extras = numpy.asarray(piece_last, dtype=numpy.int64) + 1 - runs
    run_base = numpy.concatenate([[0], numpy.cumsum(runs)])
    ext_base = numpy.concatenate([[0], numpy.cumsum(extras)])
    total_runs = int(run_base[-1])
    out = []
    for k, lab in enumerate(piece_labels):
        lab = numpy.asarray(lab, dtype=numpy.float64)
        new = numpy.where(lab < 0, -1.0, numpy.where(lab < runs[k], lab + run_base[k], lab + (total_runs - runs[k]) + ext_base[k]))
        out.append(new)
    return out


def cluster_pieces_local(buckets, pieces, ids, epsilon, m):
    """cluster the pieces `ids` of the work list on this rank's GPU -> (labels per piece (halo dropped), x-runs, final ids)"""
    arrays = []
    for i in ids:
        b, k, members, halo = pieces[i]
        a = numpy.asarray(buckets[b], dtype=numpy.int64)[:, :2]
        if members is not None:
            a = a[members]
            if halo is not None:
                a = numpy.concatenate([a, [[halo, a[0, 1] if len(a) else 0]]])      # the halo point (dist.py): never labelled
        arrays.append(a)
    labs, runs, last = cluster_buckets(arrays, epsilon, m, counts=True)
    for j, i in enumerate(ids):
        if pieces[i][3] is not None:
            labs[j] = labs[j][:-1]
    return labs, runs, last


def assemble_pieces(buckets, pieces, labels, runs, last):
    """per-piece results of ALL pieces -> list of per-bucket label arrays in signal order"""
    out = [None] * len(buckets)
    i = 0
    while i < len(pieces):
        b = pieces[i][0]
        j = i
        while j < len(pieces) and pieces[j][0] == b:
            j += 1
        if j - i == 1 and pieces[i][2] is None:
            out[b] = numpy.asarray(labels[i], dtype=numpy.float64)
        else:
            fixed = rebase_pieces(labels[i:j], runs[i:j], last[i:j])
            lab = numpy.empty(len(buckets[b]), dtype=numpy.float64)
            for k in range(i, j):
                lab[pieces[k][2]] = fixed[k - i]
            out[b] = lab
        i = j
    return out


def cluster_buckets_sharded(buckets, epsilon, m, group=None, device=None, balance=1.0, min_cut=32768):
    """Multi-GPU form of :func:`cluster_buckets` (one process per GPU, torch.distributed initialised; backend nccl = RCCL):
    the buckets — an oversized one cut into pieces at posA gaps >= epsilon (:func:`plan_pieces`) — are bin-packed onto the ranks by
    signal count, every rank clusters only its own on its GPU, and ONE variable-count all-gather (labels, then two counts per piece,
    int32 on the wire) gives every rank the full result.  Returns the same list of per-bucket label arrays on
    every rank, bit-identical to the single-GPU call."""
    import torch
    import torch.distributed as dist
    from .dist import allgatherv, shard_buckets
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pieces = plan_pieces(buckets, epsilon, world, balance, min_cut)
    psize = [len(buckets[b]) if mem is None else len(mem) for b, _, mem, _ in pieces]
    owned = shard_buckets(psize, world)
    labs, runs, last = cluster_pieces_local(buckets, pieces, owned[rank], epsilon, m)
    if (runs > 0x7fffffff).any() or (last > 0x7fffffff).any():
        raise OverflowError("cluster ids above 2^31 do not fit the int32 wire format")
    wire = numpy.concatenate([numpy.concatenate(labs) if labs else numpy.zeros(0), runs.astype(numpy.float64), last.astype(numpy.float64)])
    parts = allgatherv(torch.from_numpy(wire.astype(numpy.int32)).to(device), group)
    all_lab, all_runs, all_last = [None] * len(pieces), numpy.zeros(len(pieces), numpy.int64), numpy.zeros(len(pieces), numpy.int64)
    for r, part in enumerate(parts):
        part = part.cpu().numpy()
        k = len(owned[r])
        o = 0
        for j, i in enumerate(owned[r]):
            all_lab[i] = part[o:o + psize[i]].astype(numpy.float64)
            o += psize[i]
            all_runs[i], all_last[i] = part[len(part) - 2 * k + j], part[len(part) - k + j]
    return assemble_pieces(buckets, pieces, all_lab, all_runs, all_last)


def _mode(values):
    """Counter(values).most_common(1)[0][0] — on a tie the value that was inserted first (CPython: max() over the Counter's items in
    insertion order keeps the first maximum) — without the Counter, the heap and the key function: two calls per candidate"""
    if len(values) == 1:
        return values[0]
    counts = {}
    for v in values:
        counts[v] = counts.get(v, 0) + 1
    best, bc = None, 0
    for v, c in counts.items():
        if c > bc:
            best, bc = v, c
    return best


def _breakpoints_from_discordants(cand, is_mp):
    """:277-330 — orientation-consistent clusters take the extreme positions, others the mode."""
    A, B = cand["positions_A"], cand["positions_B"]
    revA, fwdA = A["orientation_discordants"].count("True"), A["orientation_discordants"].count("False")
    revB, fwdB = B["orientation_discordants"].count("True"), B["orientation_discordants"].count("False")
    consistent = (revA >= 5 * fwdA or revA * 5 <= fwdA) and (revB >= 5 * fwdB or revB * 5 <= fwdB)
    if not consistent:
        return _mode(A["discordants"]), _mode(B["discordants"])
    a_rev, b_rev = revA > fwdA, revB > fwdB
    if is_mp:
        pickA = max if a_rev else min
        pickB = max if b_rev else min
    else:
        pickA = min if a_rev else max
        pickB = min if b_rev else max
    return pickA(A["discordants"]), pickB(B["discordants"])


def _finish_candidates(candidates, is_mp, min_reads):
    """:256-336 — counts, breakpoints and regions of every candidate, in place"""
    for chrA in candidates:
        for chrB in candidates[chrA]:
            for cand in candidates[chrA][chrB].values():
                cand["N_discordants"] = len(cand["discordants"])
                cand["N_splits"] = len(cand["splits"])
                cand["N_contigs"] = len(cand["contigs"])
                A, B = cand["positions_A"], cand["positions_B"]
                if cand["N_splits"] and min_reads <= cand["N_splits"]:      # enough split reads: their mode (:266-268)
                    cand["posA"], cand["posB"] = _mode(A["splits"]), _mode(B["splits"])
                elif cand["N_contigs"]:
                    cand["posA"], cand["posB"] = _mode(A["contigs"]), _mode(B["contigs"])
                elif cand["N_splits"]:
                    cand["posA"], cand["posB"] = _mode(A["splits"]), _mode(B["splits"])
                else:
                    cand["posA"], cand["posB"] = _breakpoints_from_discordants(cand, is_mp)
                cand["startB"] = min(B["start"])
                cand["endB"] = max(B["end"])
                cand["startA"] = min(A["start"])
                cand["endA"] = max(A["end"])


def cluster_columns_device(posA, posB, off, epsilon, m, lab32, ctx=None):
    """int32 columns (pinned) of all buckets -> lab32[i] = cluster of signal i, on the device (``tdt_cluster_columns``: stable sort by
    posA + DBSCAN.main per bucket, :152-160)"""
    ctx = ctx or _native.default_context()
    n = len(posA)
    if not n:
        return
    lo, hi = min(int(posA.min()), int(posB.min())), max(int(posA.max()), int(posB.max()))
    _native.check(ctx.lib.tdt_cluster_columns(ctx.handle, _native.ptr(posA), _native.ptr(posB), n, _native.ptr(off), len(off) - 1, float(epsilon), int(m),
                                              hi if lo >= 0 else 0, _native.ptr(lab32), None, None))


def _native_candidates(tables, sample, is_mp, epsilon, m, min_contig, T, min_reads=None):
    """tiddit_cluster.main (:47-254) from the native signal tables of this process: the signal table is written into pinned int32
    columns by the library, clustered on the device, and the members of every candidate come back as flat arrays in the reference's
    order — the Python below runs once per CANDIDATE.  -> candidates[chrA][chrB][cluster id] for the (chrA, *) pairs the tables hold."""
    import time
    global _POOL
    t0 = time.time()
    n, nb = tables.cluster_table(is_mp, min_contig)
    if _POOL is None:
        from .hostutil import PinnedPool
        _POOL = PinnedPool()
    posA, posB, lab32 = (_POOL.take(k, n, numpy.int32) for k in ("posA", "posB", "labels"))
    off, ba, bb = tables.cluster_columns(posA, posB, nb)
    T["signal table (native, into pinned columns)"] = time.time() - t0
    t0 = time.time()
    cluster_columns_device(posA, posB, off, epsilon, m, lab32)
    T["sort + DBSCAN (device)"] = time.time() - t0
    t0 = time.time()
    g = tables.regroup(lab32)
    names = tables.names
    candidates = {}
    slots = []                                              # per bucket: candidates[chrA][chrB]
    for a, b in zip(ba.tolist(), bb.tolist()):              # header order of (chrA, chrB): the loops of :140-147
        slots.append(candidates.setdefault(names[a], {}).setdefault(names[b], {}))
    if _pycand is not None and os.environ.get("TIDDIT_PY_CANDIDATES") != "1":
        # the same dictionaries built with the CPython API (csrc/tdt_pycand.c): 15 µs of bytecode per candidate become ~3 µs
        # (with min_reads given also what _finish_candidates adds — counts, breakpoints, regions — straight from the member columns)
        extra = () if min_reads is None else (bool(is_mp), int(min_reads))
        _pycand.build(slots, g["cand"], g["names"], g["startA"], g["endA"], g["startB"], g["endB"], g["posA"], g["posB"], g["oriA"], g["oriB"], sample, *extra)
        T["regroup + breakpoints"] = time.time() - t0
        T["finished"] = min_reads is not None
        return candidates
    W = ("False", "True")
    frag = g["names"].decode().split("\n")
    sA, eA, sB, eB, pA, pB = (g[k].tolist() for k in ("startA", "endA", "startB", "endB", "posA", "posB"))
    oA, oB = [W[x] for x in g["oriA"].tolist()], [W[x] for x in g["oriB"].tolist()]
    lo = 0
    for bkt, cid, nd, ns in g["cand"].tolist():
        mid, hi = lo + nd, lo + nd + ns
        d_names, s_names = set(frag[lo:mid]), set(frag[mid:hi])
        # _new_candidate()'s dictionary (same keys, same order) written with its values in place: one literal instead of twenty
        # assignments and sixteen empty lists that twelve slices replace — this loop is the clustering stage's time at 10^4 candidates
        slots[bkt][cid] = {
            "signal_type": {}, "samples": {sample}, "sample_discordants": {sample: set(d_names)}, "sample_splits": {sample: set(s_names)},
            "sample_contigs": {sample: set()}, "N_discordants": 0, "discordants": d_names, "N_splits": 0, "splits": s_names, "N_contigs": 0,
            "contigs": set(), "n_signals": 0, "posA": 0,
            "positions_A": {"contigs": [], "splits": pA[mid:hi], "discordants": pA[lo:mid], "orientation_contigs": [],
                            "orientation_splits": oA[mid:hi], "orientation_discordants": oA[lo:mid], "start": sA[lo:hi], "end": eA[lo:hi]},
            "start_A": 0, "end_A": 0, "posB": 0,
            "positions_B": {"contigs": [], "splits": pB[mid:hi], "discordants": pB[lo:mid], "orientation_contigs": [],
                            "orientation_splits": oB[mid:hi], "orientation_discordants": oB[lo:mid], "start": sB[lo:hi], "end": eB[lo:hi]},
            "start_B": 0, "end_B": 0}
        lo = hi
    T["regroup + breakpoints"] = time.time() - t0
    return candidates


def _handed_over(prefix, chromosomes, contig_length, samples, min_contig, skip_assembly):
    """the native tables tiddit_signal.main of THIS process wrote the sample's files from, if they describe the same contigs"""
    if not skip_assembly or len(samples) != 1:
        return None, None
    from . import tiddit_signal
    paths = ("{}_tiddit/discordants_{}.tab".format(prefix, samples[0]), "{}_tiddit/splits_{}.tab".format(prefix, samples[0]))
    tables = tiddit_signal.written_tables(*paths)
    if tables is None or tables.names != list(chromosomes) or tables.min_contig > min_contig:
        return None, None
    if any(contig_length.get(n) != ln for n, ln in zip(tables.names, tables.lengths)):
        return None, None
    return tables, tiddit_signal.table_owners(*paths)


def _is_sharded():
    try:
        import torch.distributed as _dist
        import os
        return _dist.is_available() and _dist.is_initialized() and (_dist.get_world_size() > 1 or os.environ.get("TIDDIT_FORCE_DIST") == "1")
    except ImportError:
        return False


def _agree_native(native):
    """the native-vs-text choice of an N-rank job must come out the same on every rank (the two branches run different collectives):
    one int, all-reduced with MIN"""
    import torch
    import torch.distributed as _dist
    from .dist import _wire_device
    flag = torch.tensor([1 if native else 0], dtype=torch.int32, device=_wire_device(None))
    _dist.all_reduce(flag, op=_dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def _main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads, root_only=False,
          sharded=False):
    """`sharded` is the caller's statement, not a look at global state: plain main() is ONE process's call and never enters a
    collective, whatever torch.distributed group its host application has initialised; main_sharded() is every rank's call."""
    import sys
    import time
    STAGE_SECONDS.clear()
    tables, owner = _handed_over(prefix, chromosomes, contig_length, samples, min_contig, skip_assembly)
    native = tables is not None and (owner is not None) == sharded
    if sharded:
        native = _agree_native(native)
    if not native:
        # the text is about to be parsed: a writer thread of THIS process (tiddit_signal.BACKGROUND_WRITES) may still be placing the
        # blocks of exactly these files, which already exist at their final size
        ts = sys.modules.get(__package__ + ".tiddit_signal")
        if ts is not None:
            ts.finish_writes()
    if native:
        candidates = _native_candidates(tables, samples[0], is_mp, epsilon, m, min_contig, STAGE_SECONDS, min_reads=min_reads)
        if not STAGE_SECONDS.pop("finished", False):          # (the Python loop built them: counts, breakpoints and regions follow)
            t0 = time.time()
            _finish_candidates(candidates, is_mp, min_reads)
            STAGE_SECONDS["regroup + breakpoints"] += time.time() - t0
        if sharded:
            # every rank holds the candidates of the chrA it owns: rank 0 puts them together in header order (the order of :140-147)
            import pickle
            import torch.distributed as _dist
            from .dist import gather_bytes
            t1 = time.time()
            parts = gather_bytes(pickle.dumps(candidates, protocol=4), 0)
            if _dist.get_rank() == 0:
                mine = [pickle.loads(p) for p in parts]
                candidates = {}
                for t, chrA in enumerate(chromosomes):
                    part = mine[int(owner[t])]
                    if chrA in part:
                        candidates[chrA] = part[chrA]
            elif root_only:
                candidates = None
            STAGE_SECONDS["candidates to rank 0"] = time.time() - t1
        return candidates
    t0 = time.time()
    signals, positions = _read_signals(prefix, samples, contig_length, is_mp, min_contig, skip_assembly)
    STAGE_SECONDS["parse .tab"] = time.time() - t0
    t0 = time.time()

    order = [(a, b) for a in chromosomes if a in positions for b in chromosomes if b in positions[a]]
    bucket_arrays = [numpy.array(positions[a][b], dtype=numpy.int64).reshape(-1, 3) for a, b in order]
    if sharded:
        import os
        import torch.distributed as _dist
        labels = cluster_buckets_sharded(bucket_arrays, epsilon, m, min_cut=int(os.environ.get("TIDDIT_CLUSTER_MIN_CUT", "32768")))
    else:
        labels = cluster_buckets(bucket_arrays, epsilon, m)

    STAGE_SECONDS["sort + DBSCAN (device)"] = time.time() - t0
    if sharded and root_only and _dist.get_rank() != 0:
        return None                                      # the labels are on every rank; the candidates dictionary is rank 0's job
    t0 = time.time()
    candidates = {}
    for chrA in chromosomes:           # candidates[chrA] exists for every chrA that has signals (:141-145)
        if chrA in positions:
            candidates[chrA] = {}
    for (chrA, chrB), lab in zip(order, labels):
        bucket = candidates[chrA].setdefault(chrB, {})
        recs = signals[chrA][chrB]
        n_sig = len(recs)
        n_ctg_clusters = 0
        for rec, cid in zip(recs, lab.astype(numpy.int64).tolist()):         # signal-index order == file order inside the bucket (:160)
            if cid == -1:
                lone_contig = chrA == chrB and rec[2] == "A" and (int(rec[5]) - int(rec[3])) < max_ins_len * 2
                if not lone_contig:
                    continue
                cid = n_sig + n_ctg_clusters        # unclustered assembly contigs become their own candidate (:166-168)
                n_ctg_clusters += 1
            cand = bucket.get(cid)
            if cand is None:
                cand = bucket[cid] = _new_candidate()
            qname, sample, kind = rec[0], rec[1], rec[2]
            if sample not in cand["samples"]:
                cand["sample_discordants"][sample] = set([])
                cand["sample_splits"][sample] = set([])
                cand["sample_contigs"][sample] = set([])
            cand["samples"].add(sample)
            cand["positions_A"]["start"].append(rec[8])
            cand["positions_A"]["end"].append(rec[9])
            cand["positions_B"]["start"].append(rec[10])
            cand["positions_B"]["end"].append(rec[11])
            name = _KIND[kind]
            cand[name].add(qname)
            cand["positions_A"][name].append(int(rec[3]))
            cand["positions_A"]["orientation_" + name].append(rec[4])
            cand["positions_B"][name].append(int(rec[5]))
            cand["positions_B"]["orientation_" + name].append(rec[6])
            cand["sample_" + name][sample].add(qname)
    _finish_candidates(candidates, is_mp, min_reads)
    STAGE_SECONDS["regroup + breakpoints"] = time.time() - t0
    return candidates


def main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads):
    """``tiddit_cluster.main`` (tiddit_cluster.pyx:39-336): .tab files -> candidates dictionary.  (The collector is off while the
    tables are built: hostutil.quiet_gc.)"""
    with quiet_gc():
        return _main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads)


def main_sharded(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads):
    """:func:`main` for one process per GPU (torch.distributed initialised).  After tiddit_signal.main_sharded of the same job every
    rank holds the native tables of the chrA it owns: it clusters THEIR buckets on its GPU and regroups THEIR candidates; rank 0
    receives the finished candidates (pickled, ~10^4 of them) and returns the dictionary (the other ranks return None).  On files
    from elsewhere every rank parses the text, the buckets are clustered where :func:`cluster_buckets_sharded` puts them and rank 0
    regroups."""
    with quiet_gc():
        return _main(prefix, chromosomes, contig_length, samples, is_mp, epsilon, m, max_ins_len, min_contig, skip_assembly, min_reads, root_only=True,
                     sharded=_is_sharded())
