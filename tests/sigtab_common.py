"""Host stand-ins for the two device steps of the `tiddit --sv` signal path, so that the native signal tables (csrc/tdt_sigtab.hip,
host code) and everything built on them can be tested without a GPU:
  * select_host — what ``tdt_signal_scan`` + ``tdt_signal_scan_result`` return for a decoded batch (the per-read chain of
    tiddit_signal.worker, tiddit_signal.pyx:171-221, as numpy predicates — the same expressions as the host-ingest branch of
    tiddit_signal._scan — and the selected reads' 28-byte field records + raw BAM records);
  * oracle_labels — what ``tdt_cluster_columns`` returns (stable sort by posA + the oracle's DBSCAN.main per bucket).
Test infrastructure only."""
import numpy as np

from tiddit_amd import tiddit_signal


def select_host(b, big, min_q, max_ins, min_anchor_len, min_clip_len):
    """-> (meta, raw_end, raw) of the reads of host batch `b` with an action bit set, in file order"""
    tid = b.tid
    flag = b.flag.astype(np.int32)
    placed = (tid >= 0) & (tid < len(big))
    ok_contig = np.zeros(len(tid), dtype=bool)
    ok_contig[placed] = np.asarray(big, dtype=bool)[tid[placed]]
    primary = ok_contig & ((flag & 0x404) == 0) & ((flag & 0x900) == 0) & (b.mapq >= min_q)
    same = b.mate_tid == tid
    isz = np.abs(b.tlen.astype(np.int64))
    f_op, f_len = b.cigar_first & 0xf, (b.cigar_first >> 4).astype(np.int64)
    l_op, l_len = b.cigar_last & 0xf, (b.cigar_last >> 4).astype(np.int64)
    has_cigar = b.cigar_first != 0xffffffff
    left = (f_op == 4) & (f_len > min_clip_len) & (l_op == 0) & (l_len > min_anchor_len)
    right = (l_op == 4) & (l_len > min_clip_len) & (f_op == 0) & (f_len > min_anchor_len)
    act = np.zeros(len(tid), dtype=np.uint8)
    act[primary & (isz < max_ins) & same & has_cigar & (left | right)] |= 2
    act[primary & (b.sa_off >= 0)] |= 4
    act[primary & ((flag & 0x8) == 0) & ((flag & 0x1) != 0) & (b.mate_tid >= 0) & ((isz > max_ins) | ~same)] |= 8
    idx = np.flatnonzero(act)
    meta = np.zeros(len(idx), dtype=tiddit_signal._META)
    meta["idx"] = idx
    for k in ("tid", "pos", "end", "mate_tid", "flag"):
        meta[k] = getattr(b, k)[idx]
    meta["action"] = act[idx]
    off = b.rec_off[idx].astype(np.int64)
    sa = b.sa_off[idx]
    meta["sa_rel"] = np.where(sa >= 0, sa - off, -1)
    raw = np.asarray(b.raw)
    size = np.array([int(raw[o:o + 4].view("<i4")[0]) + 4 for o in off], dtype=np.int64) if len(idx) else np.zeros(0, dtype=np.int64)
    raw_end = np.cumsum(size).astype(np.uint32)
    out = np.concatenate([raw[o:o + s] for o, s in zip(off, size)]) if len(idx) else np.zeros(0, dtype=np.uint8)
    return meta, raw_end, np.ascontiguousarray(out)


class HostSel:
    """quacks like tiddit_signal.SelectedReads for the literal SA_analysis fallback"""

    def __init__(self, meta, raw_end, raw):
        from tiddit_amd import bamio
        self.meta, self.raw_end, self.raw = meta, raw_end, raw
        self.raw_bytes = raw.tobytes()
        self.tid, self.pos, self.end, self.flag = meta["tid"], meta["pos"], meta["end"], meta["flag"]
        self.rec_off = np.concatenate([[0], raw_end[:-1]]).astype(np.uint64) if len(raw_end) else np.zeros(0, dtype=np.uint64)
        self.sa_off = np.where(meta["sa_rel"] >= 0, self.rec_off.astype(np.int64) + meta["sa_rel"], -1)
        self.record = lambda i: bamio.RecordView(self, i)


def fill_tables(tables, batches, names, big, min_q, max_ins, min_anchor_len, min_clip_len):
    """every batch's selected reads into `tables`, as tiddit_signal._scan's row thread does; -> selected reads"""
    n = 0
    for b in batches:
        meta, raw_end, raw = select_host(b, big, min_q, int(max_ins), min_anchor_len, min_clip_len)
        sel = HostSel(meta, raw_end, raw)
        tables.add(meta, raw_end, raw, min_q,
                   literal=lambda k, sel=sel: tiddit_signal.SA_analysis(tiddit_signal._ReadProxy(sel, k), min_q, "SA", names[int(sel.tid[k])]))
        n += len(meta)
    return n


def oracle_labels(posA, posB, off, epsilon, m, lab32, ctx=None):
    """drop-in for tiddit_cluster.cluster_columns_device on a machine without a GPU: per bucket the stable sort by posA and the
    oracle's DBSCAN.main (tiddit_cluster.pyx:152-160), labels back in signal order"""
    import oracle
    for b in range(len(off) - 1):
        lo, hi = int(off[b]), int(off[b + 1])
        pts = np.stack([posA[lo:hi], posB[lo:hi]], axis=1).astype(np.int64)
        order = np.argsort(pts[:, 0], kind="stable")
        lab = oracle.dbscan_main(pts[order], epsilon, m)
        out = np.empty(hi - lo, dtype=np.int32)
        out[order] = np.asarray(lab).astype(np.int32)
        lab32[lo:hi] = out
