"""The signal tables of ``tiddit --sv`` behind the C ABI (``tdt_sigtab_*``, csrc/tdt_sigtab.hip): the discordant / split / clipped
reads that ``tdt_signal_scan`` selected go from the device batch into native tables — no Python object per read — which then give
the bytes of ``discordants_{sample}.tab`` / ``splits_{sample}.tab`` / the clip FASTA files (tiddit_signal.pyx:246-332), the
``(posA, posB)`` columns of every ``(chrA, chrB)`` bucket (tiddit_cluster.pyx:47-105) and, with the labels, the members of every
candidate (tiddit_cluster.pyx:156-254).  ``tiddit_signal`` fills a table, ``tiddit_cluster`` in the same process takes it over."""
import ctypes

import numpy

from . import _native

D_ROW = numpy.dtype([("tid", "<i4"), ("mate", "<i4"), ("start", "<i4"), ("end", "<i4"), ("name_off", "<u8"), ("name_len", "<u2"), ("rev", "u1"),
                     ("pad", "u1"), ("pad2", "<u4")])
S_ROW = numpy.dtype([("tid", "<i4"), ("a", "<i4"), ("b", "<i4"), ("name_len", "<u2"), ("rev", "u1"), ("sa_minus", "u1"), ("name_off", "<u8"),
                     ("other_off", "<u8"), ("other_len", "<u4"), ("pad", "<u4"), ("f", "<i8", (6,))])
assert D_ROW.itemsize == 32 and S_ROW.itemsize == 88


class SignalTables:
    def __init__(self, names, lengths, min_contig, lib=None):
        self.lib = lib or _native.load()
        self.names = list(names)
        self.lengths = [int(x) for x in lengths]
        self.min_contig = int(min_contig)
        blob = b"".join(n.encode() + b"\0" for n in self.names)
        ln = numpy.asarray(self.lengths, dtype=numpy.int64)
        h = ctypes.c_void_p()
        _native.check(self.lib.tdt_sigtab_create(blob, _native.ptr(ln), len(self.names), self.min_contig, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h is not None:
            self.lib.tdt_sigtab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == -7:                                   # TDT_E_KEY: where the reference's dictionaries raise KeyError
            raise KeyError(self.lib.tdt_last_error().decode(errors="replace"))
        _native.check(rc)

    # ---- filling ----------------------------------------------------------------------------------------------------
    def add(self, meta, raw_end, raw, min_q, literal=None):
        """the selected reads of one batch (arrays of tdt_signal_scan_result) in file order.  ``literal(k)`` -> the split row of
        selected read k by the literal SA_analysis (a list of 11, or something empty), for the SA tags the C parser does not take."""
        n = len(meta)
        stopped = ctypes.c_size_t(0)
        resume = 0
        while True:
            self._check(self.lib.tdt_sigtab_add(self._h, _native.ptr(meta), _native.ptr(raw_end), _native.ptr(raw), n, len(raw), int(min_q), resume,
                                                ctypes.byref(stopped)))
            k = stopped.value
            if k >= n:
                return
            if literal is None:
                raise ValueError("an SA tag outside the C parser's domain and no literal fallback")
            row = literal(k)
            if row:
                self.add_split_row(int(meta["tid"][k]), row)
            resume = k + 1
            if resume >= n:
                # (the batch's rows before the stop are merged; nothing is left)
                return

    def add_split_row(self, tid, row):
        """row = [chrA, chrB, name, split_pos, is_reverse, sa_split, sa_minus, startA, endA, startB, endB] (SA_analysis, :138-142)"""
        six = numpy.array([row[3], row[5], row[7], row[8], row[9], row[10]], dtype=numpy.int64)
        self._check(self.lib.tdt_sigtab_add_split_row(self._h, int(tid), row[0].encode(), row[1].encode(), row[2].encode(), _native.ptr(six),
                                                     int(bool(row[4])), int(bool(row[6]))))

    def add_clips(self, tid, data):
        buf = numpy.frombuffer(data, dtype=numpy.uint8)
        self._check(self.lib.tdt_sigtab_add_clips(self._h, int(tid), _native.ptr(buf) if len(buf) else None, len(buf)))

    def import_rows(self, blob):
        blob = numpy.ascontiguousarray(numpy.frombuffer(blob, dtype=numpy.uint8))
        self._check(self.lib.tdt_sigtab_import(self._h, _native.ptr(blob), len(blob)))

    # ---- reading ----------------------------------------------------------------------------------------------------
    def stats(self):
        out = numpy.zeros(8, dtype=numpy.int64)
        self._check(self.lib.tdt_sigtab_stats(self._h, _native.ptr(out)))
        return dict(discordant_rows=int(out[0]), split_rows=int(out[1]), pairs=int(out[2]), discordants_in_order=bool(out[3]),
                    splits_in_order=bool(out[4]), name_bytes=int(out[5]), clip_bytes=int(out[6]))

    def clips(self, tid):
        p, n = ctypes.c_void_p(), ctypes.c_size_t(0)
        self._check(self.lib.tdt_sigtab_clips(self._h, int(tid), ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value) if n.value else b""

    def export_rows(self, owner=None, dest=0):
        """the row log as one uint8 array (rows whose chrA is owned by rank `dest`; owner None: every row)"""
        own = None if owner is None else numpy.ascontiguousarray(owner, dtype=numpy.int32)
        need = ctypes.c_size_t(0)
        self._check(self.lib.tdt_sigtab_export(self._h, _native.ptr(own), int(dest), None, 0, ctypes.byref(need)))
        out = numpy.empty(need.value, dtype=numpy.uint8)
        self._check(self.lib.tdt_sigtab_export(self._h, _native.ptr(own), int(dest), _native.ptr(out), need.value, ctypes.byref(need)))
        return out

    def rows(self):
        """-> (data, splits): per contig NAME the rows worker() returns (tiddit_signal.pyx:214-221 and SA_analysis :138-142), in file
        order — Python lists, for callers of the reference's per-contig interface; the job itself never builds them."""
        blob = self.export_rows()
        nd, ns, nb = (int(x) for x in blob[8:32].view("<u8"))
        o = 32
        d = blob[o:o + nd * 32].view(D_ROW)
        o += nd * 32
        s = blob[o:o + ns * 88].view(S_ROW)
        o += ns * 88
        nm = blob[o:o + nb].tobytes()
        names = self.names
        data = {n: [] for n in names}
        splits = {n: [] for n in names}
        for tid, mate, start, end, off, ln, rev in zip(d["tid"].tolist(), d["mate"].tolist(), d["start"].tolist(), d["end"].tolist(),
                                                       d["name_off"].tolist(), d["name_len"].tolist(), d["rev"].tolist()):
            chrom, other = names[tid], names[mate]
            chrA, chrB = (other, chrom) if other < chrom else (chrom, other)
            data[chrom].append([chrA, chrB, nm[off:off + ln].decode(), start, end, bool(rev), chrom])
        for tid, a, b, off, ln, ooff, oln, rev, sam, f in zip(s["tid"].tolist(), s["a"].tolist(), s["b"].tolist(), s["name_off"].tolist(),
                                                              s["name_len"].tolist(), s["other_off"].tolist(), s["other_len"].tolist(),
                                                              s["rev"].tolist(), s["sa_minus"].tolist(), s["f"].tolist()):
            other = nm[ooff:ooff + oln].decode()
            chrA = names[a] if a >= 0 else other
            chrB = names[b] if b >= 0 else other
            splits[names[tid]].append([chrA, chrB, nm[off:off + ln].decode(), f[0], bool(rev), f[1], bool(sam), f[2], f[3], f[4], f[5]])
        return data, splits

    def text(self, kind):
        """-> (bytes of the table, segments int64[n, 5]: chrA id, chrB id, offset, length, rows); kind 0 discordants, 1 splits"""
        nseg = (ctypes.c_size_t(0), ctypes.c_size_t(0))
        self._check(self.lib.tdt_sigtab_format(self._h, ctypes.byref(nseg[0]), ctypes.byref(nseg[1])))
        seg = numpy.zeros((nseg[kind].value, 5), dtype=numpy.int64)
        p, n = ctypes.c_void_p(), ctypes.c_size_t(0)
        self._check(self.lib.tdt_sigtab_text(self._h, int(kind), ctypes.byref(p), ctypes.byref(n), _native.ptr(seg)))
        return (ctypes.string_at(p, n.value) if n.value else b""), seg

    def sizes(self, what):
        """bytes per contig id of one output: what 0 = the discordant rows with that chrA, 1 = the split rows, 2 = the contig's clip FASTA"""
        out = numpy.zeros(len(self.names), dtype=numpy.int64)
        self._check(self.lib.tdt_sigtab_sizes(self._h, int(what), _native.ptr(out)))
        return out

    def pwrite(self, what, contig, fd, offset):
        self._check(self.lib.tdt_sigtab_pwrite(self._h, int(what), int(contig), int(fd), int(offset)))

    # ---- clustering -------------------------------------------------------------------------------------------------
    def cluster_table(self, is_mp, min_contig):
        n, nb = ctypes.c_size_t(0), ctypes.c_int(0)
        self._check(self.lib.tdt_sigtab_cluster_table(self._h, int(bool(is_mp)), int(min_contig), ctypes.byref(n), ctypes.byref(nb)))
        return n.value, nb.value

    def cluster_columns(self, posA, posB, nb):
        """fills posA / posB (int32 arrays of n_signals, e.g. pinned) -> (bucket_off int64[nb + 1], chrA ids, chrB ids)"""
        off = numpy.zeros(nb + 1, dtype=numpy.int64)
        a, b = numpy.zeros(nb, dtype=numpy.int32), numpy.zeros(nb, dtype=numpy.int32)
        self._check(self.lib.tdt_sigtab_cluster_columns(self._h, _native.ptr(posA), _native.ptr(posB), _native.ptr(off), _native.ptr(a), _native.ptr(b)))
        return off, a, b

    def regroup(self, labels):
        """labels int32[n_signals] (-1 = noise) -> dict of arrays describing every candidate and its members (see tdt_sigtab_regroup_result)"""
        labels = numpy.ascontiguousarray(labels, dtype=numpy.int32)
        nc, nm, nbytes = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self.lib.tdt_sigtab_regroup(self._h, _native.ptr(labels), ctypes.byref(nc), ctypes.byref(nm), ctypes.byref(nbytes)))
        m = nm.value
        cand = numpy.zeros((nc.value, 4), dtype=numpy.int32)
        i32 = [numpy.zeros(m, dtype=numpy.int32) for _ in range(6)]
        u8 = [numpy.zeros(m, dtype=numpy.uint8) for _ in range(2)]
        names = numpy.zeros(max(nbytes.value, 1), dtype=numpy.uint8)
        self._check(self.lib.tdt_sigtab_regroup_result(self._h, _native.ptr(cand), *[_native.ptr(x) for x in i32], *[_native.ptr(x) for x in u8],
                                                      _native.ptr(names)))
        return dict(cand=cand, startA=i32[0], endA=i32[1], startB=i32[2], endB=i32[3], posA=i32[4], posB=i32[5], oriA=u8[0], oriB=u8[1],
                    names=names[:nbytes.value].tobytes())
