"""Minimal BGZF/BAM reader + writer (the reference reads alignments through pysam/htslib, which is not
a dependency here).

``BamReader.batches()`` inflates BGZF blocks on the native host thread pool (``tdt_bgzf_inflate``, csrc/tdt_bgzf.hip)
and hands whole-record pieces to the C decoder ``tdt_bam_decode`` (csrc/tdt_bam.hip), yielding struct-of-arrays batches: the packed
start/end/mapq/flag arrays go straight to the coverage kernel, the mate fields drive the discordant-pair
predicate, and the few reads that need string work (SA tags, clipped sequences, names) are pulled out of
the raw record bytes with :class:`RecordView`.  ``BamWriter`` produces small coordinate-sorted BAMs for
tests and synthetic configs.  Semantics follow the SAM/BAM spec v1; ``end`` is htslib ``bam_endpos``.
"""
import ctypes
import os
import struct
import time
import zlib

import numpy as np

from . import _native

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
CIGAR_OPS = "MIDNSHP=X"
_SEQ_CODES = "=ACMGRSVTWYHKDBN"


# ------------------------------------------------------------------------------------- BGZF
def bgzf_blocks(f):
    """yield the uncompressed payload of every BGZF block of an open binary file"""
    while True:
        hdr = f.read(12)
        if len(hdr) < 12:
            return
        if hdr[0] != 0x1f or hdr[1] != 0x8b or not (hdr[3] & 4):
            raise ValueError("not a BGZF file")
        xlen = struct.unpack_from("<H", hdr, 10)[0]
        extra = f.read(xlen)
        bsize = None
        o = 0
        while o + 4 <= xlen:
            si1, si2, slen = extra[o], extra[o + 1], struct.unpack_from("<H", extra, o + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", extra, o + 4)[0]
            o += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC field")
        cdata = f.read(bsize - xlen - 19)
        f.read(8)  # crc32, isize
        yield zlib.decompress(cdata, -15) if cdata else b""


def bgzf_blocks_parallel(f, threads=None, batch=256):
    """Same stream as :func:`bgzf_blocks`, inflating `batch` blocks at a time on a thread pool (zlib releases the
    GIL): BGZF inflate is the end-to-end bottleneck of `tiddit --cov`/`--sv` once the histograms run on the GPU."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(16, os.cpu_count() or 1)
    if threads <= 1:
        yield from bgzf_blocks(f)
        return

    def raw_blocks():
        while True:
            hdr = f.read(12)
            if len(hdr) < 12:
                return
            if hdr[0] != 0x1f or hdr[1] != 0x8b or not (hdr[3] & 4):
                raise ValueError("not a BGZF file")
            xlen = struct.unpack_from("<H", hdr, 10)[0]
            extra = f.read(xlen)
            bsize = None
            o = 0
            while o + 4 <= xlen:
                if extra[o] == 66 and extra[o + 1] == 67:
                    bsize = struct.unpack_from("<H", extra, o + 4)[0]
                o += 4 + struct.unpack_from("<H", extra, o + 2)[0]
            if bsize is None:
                raise ValueError("BGZF block without BC field")
            cdata = f.read(bsize - xlen - 19)
            f.read(8)
            yield cdata

    inflate = lambda c: zlib.decompress(c, -15) if c else b""
    with ThreadPoolExecutor(threads) as pool:
        group = []
        for c in raw_blocks():
            group.append(c)
            if len(group) == batch:
                yield from pool.map(inflate, group)
                group = []
        if group:
            yield from pool.map(inflate, group)


def _bgzf_block(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


# ------------------------------------------------------------------------------------ reader
class RecordView:
    """Lazy string-level access to ONE record of a batch (name, CIGAR, sequence, aux tags)."""

    def __init__(self, batch, i):
        self.batch, self.i = batch, i
        self.raw = getattr(batch, "raw_bytes", None) or batch.raw       # (a bytes copy, where the batch keeps one: cheaper to slice)
        self.off = int(batch.rec_off[i]) + 4

    def _u(self, fmt, o):
        return struct.unpack_from(fmt, self.raw, self.off + o)[0]

    @property
    def query_name(self):
        l = self._u("<B", 8)
        return bytes(self.raw[self.off + 32:self.off + 32 + l - 1]).decode()

    @property
    def cigartuples(self):
        l_name, n = self._u("<B", 8), self._u("<H", 12)
        words = struct.unpack_from("<%dI" % n, self.raw, self.off + 32 + l_name)
        return [(w & 0xf, w >> 4) for w in words]

    @property
    def query_sequence(self):
        l_name, n, lseq = self._u("<B", 8), self._u("<H", 12), self._u("<i", 16)
        o = self.off + 32 + l_name + 4 * n
        packed = np.frombuffer(self.raw, dtype=np.uint8, count=(lseq + 1) // 2, offset=o)
        codes = np.empty(2 * len(packed), dtype=np.uint8)
        codes[0::2] = packed >> 4
        codes[1::2] = packed & 0xf
        lut = np.frombuffer(_SEQ_CODES.encode(), dtype=np.uint8)
        return lut[codes[:lseq]].tobytes().decode()

    def get_tag_sa(self):
        o = int(self.batch.sa_off[self.i])
        if o < 0:
            raise KeyError("SA")
        raw = self.raw
        if isinstance(raw, np.ndarray):
            w = 256
            while True:
                z = np.flatnonzero(raw[o:o + w] == 0)
                if len(z) or o + w >= len(raw):
                    break
                w *= 4
            e = o + int(z[0])
        else:
            e = raw.index(b"\x00", o)
        return bytes(raw[o:e]).decode()


class Batch:
    """Struct-of-arrays view of consecutive BAM records (numpy arrays of equal length)."""
    __slots__ = ("raw", "tid", "pos", "end", "mapq", "flag", "mate_tid", "mate_pos", "tlen", "l_seq", "cigar_first",
                 "cigar_last", "rec_off", "sa_off")

    def __len__(self):
        return len(self.pos)

    def record(self, i):
        return RecordView(self, i)


def inflate_pieces(f, chunk=32 << 20, max_out=192 << 20, gap=1 << 20, depth=2):
    """The uncompressed byte stream of BGZF file object `f` in large pieces: uint8 arrays whose first `gap` bytes are
    free (the caller parks the previous piece's partial record there).  The file is read `chunk` compressed bytes at a
    time, whole blocks are found by ``tdt_bgzf_scan`` and inflated on the host thread pool by ``tdt_bgzf_inflate``
    (CRC32/ISIZE checked); a helper thread keeps up to `depth` pieces ahead of the consumer."""
    import queue
    import threading
    lib = _native.load()
    q = queue.Queue(maxsize=depth)

    def produce():
        try:
            comp = np.empty(chunk + (1 << 17), dtype=np.uint8)
            have, eof = 0, False
            step = min(chunk, 1 << 20)                              # a small first piece: the header is wanted right away
            while True:
                if not eof:
                    got = f.readinto(memoryview(comp)[have:have + step])
                    step = chunk
                    eof = not got
                    have += got or 0
                if have == 0:
                    break
                nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
                _native.check(lib.tdt_bgzf_scan(_native.ptr(comp), have, max_out, ctypes.byref(nb), ctypes.byref(consumed),
                                                ctypes.byref(produced)))
                if nb.value == 0:
                    if eof:
                        raise ValueError("truncated BGZF block at end of file")
                    if have >= len(comp) - chunk // 2 - 1:
                        raise ValueError("BGZF block larger than the read window")
                    continue
                out = np.empty(gap + produced.value, dtype=np.uint8)
                _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(out[gap:]), produced.value, 0))
                rest = have - consumed.value
                comp[:rest] = comp[consumed.value:have].copy()
                have = rest
                if len(comp) < have + chunk:
                    comp = np.concatenate([comp[:have], np.empty(chunk + (1 << 17), dtype=np.uint8)])
                q.put(out)
            q.put(None)
        except BaseException as e:      # hand the failure to the consumer
            q.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    while True:
        item = q.get()
        if item is None:
            break
        if isinstance(item, BaseException):
            raise item
        yield item
    th.join()


class BamReader:
    GAP = 1 << 20

    def __init__(self, path, batch_bytes=None):
        self.path = path
        self._f = open(path, "rb", buffering=0)
        kw = {} if not batch_bytes else {"chunk": max(1 << 16, batch_bytes // 2), "max_out": max(1 << 17, batch_bytes)}
        self._pieces = inflate_pieces(self._f, gap=self.GAP, **kw)
        self._buf = bytearray()
        self._read_header()

    def _need(self, n):
        while len(self._buf) < n:
            try:
                self._buf += memoryview(next(self._pieces))[self.GAP:]
            except StopIteration:
                return False
        return True

    def _take(self, n):
        if not self._need(n):
            raise ValueError("truncated BAM")
        self.header_bytes = getattr(self, "header_bytes", 0) + n      # after _read_header: inflated bytes before record 0
        out = bytes(self._buf[:n])
        del self._buf[:n]
        return out

    def _read_header(self):
        if self._take(4) != b"BAM\x01":
            raise ValueError("not a BAM file")
        l_text = struct.unpack("<i", self._take(4))[0]
        self.text = self._take(l_text).split(b"\x00")[0].decode()
        n_ref = struct.unpack("<i", self._take(4))[0]
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._take(4))[0]
            self.references.append(self._take(l_name)[:-1].decode())
            self.lengths.append(struct.unpack("<i", self._take(4))[0])
        # pysam-style header dict (what create_coverage / the CLI read: header["SQ"][i]["SN"/"LN"], ["RG"][0]["SM"])
        self.header = {"SQ": [{"SN": n, "LN": l} for n, l in zip(self.references, self.lengths)]}
        for line in self.text.split("\n"):
            if line.startswith("@RG"):
                rg = {}
                for field in line.split("\t")[1:]:
                    if ":" in field:
                        k, v = field.split(":", 1)
                        rg[k] = v
                self.header.setdefault("RG", []).append(rg)

    def batches(self):
        """yield :class:`Batch` objects of whole records, one per inflated piece"""
        lib = _native.load()
        pending = np.frombuffer(bytes(self._buf), dtype=np.uint8)        # what followed the header in its piece
        self._buf = bytearray()
        first, done = len(pending) > 0, False
        while not done:
            if first:
                raw, first = pending, False
            else:
                piece = next(self._pieces, None)
                if piece is None:
                    if len(pending):
                        raise ValueError("truncated BAM record at end of file")
                    break
                if len(pending) <= self.GAP:                              # park the partial record in the piece's gap
                    piece[self.GAP - len(pending):self.GAP] = pending
                    raw = piece[self.GAP - len(pending):]
                else:
                    raw = np.concatenate([pending, piece[self.GAP:]])
            cap = len(raw) // 36 + 1
            b = Batch()
            b.raw = raw
            arrs = {"tid": np.int32, "pos": np.int32, "end": np.int32, "mapq": np.uint8, "flag": np.uint16, "mate_tid": np.int32,
                    "mate_pos": np.int32, "tlen": np.int32, "l_seq": np.int32, "cigar_first": np.uint32, "cigar_last": np.uint32,
                    "rec_off": np.uint64, "sa_off": np.int64}
            for k, dt in arrs.items():
                setattr(b, k, np.empty(cap, dtype=dt))
            consumed, nrec = ctypes.c_size_t(0), ctypes.c_size_t(0)
            _native.check(lib.tdt_bam_decode(_native.ptr(raw), len(raw), cap, ctypes.byref(consumed), ctypes.byref(nrec),
                                             *[_native.ptr(getattr(b, k)) for k in arrs]))
            n = nrec.value
            for k in arrs:
                setattr(b, k, getattr(b, k)[:n])
            pending = raw[consumed.value:]
            if len(pending) <= self.GAP:
                pending = pending.copy()
            if n:
                yield b

    def close(self):
        self._f.close()


def find_block_start(path, offset):
    """File offset of the first BGZF block that starts at or after `offset` (how a shard of a BAM finds its feet):
    the gzip/BGZF magic with a BC subfield whose BSIZE leads to another block header (or to the end of the file)."""
    import os
    fsize = os.path.getsize(path)
    if offset <= 0:
        return 0
    if offset >= fsize:
        return fsize

    def header_at(buf, p):
        if p + 18 > len(buf) or buf[p:p + 4] != b"\x1f\x8b\x08\x04":
            return None
        xlen = struct.unpack_from("<H", buf, p + 10)[0]
        o = p + 12
        while o + 4 <= p + 12 + xlen and o + 4 <= len(buf):
            slen = struct.unpack_from("<H", buf, o + 2)[0]
            if buf[o] == 66 and buf[o + 1] == 67 and slen == 2 and o + 6 <= len(buf):
                return struct.unpack_from("<H", buf, o + 4)[0] + 1
            o += 4 + slen
        return None

    with open(path, "rb") as f:
        f.seek(offset)
        buf = f.read(4 << 16)
    p = buf.find(b"\x1f\x8b\x08\x04")
    while p >= 0:
        q, ok = p, True
        for _ in range(3):                                    # three consecutive headers (or a clean end of file)
            bs = header_at(buf, q)
            if bs is None:
                ok = offset + q == fsize
                break
            q += bs
            if offset + q == fsize:
                break
            if q + 18 > len(buf):
                break
        if ok:
            return offset + p
        p = buf.find(b"\x1f\x8b\x08\x04", p + 1)
    raise ValueError("no BGZF block found after offset %d" % offset)


# ------------------------------------------------------------------------------------ device reader
_FIELDS = (("tid", np.int32), ("pos", np.int32), ("end", np.int32), ("mapq", np.uint8), ("flag", np.uint16), ("mate_tid", np.int32),
           ("mate_pos", np.int32), ("tlen", np.int32), ("l_seq", np.int32), ("cigar_first", np.uint32), ("cigar_last", np.uint32),
           ("rec_off", np.uint64), ("sa_off", np.int64))


class DeviceBatch:
    """One ingest batch resident in HBM (csrc/tdt_ingest.hip).  ``dev[name]`` is the device pointer of a field array,
    ``runs`` the per-contig record ranges ``[(tid, lo, hi)]``; reading a field attribute (``b.pos``, ``b.raw`` ...)
    copies that array to the host once.  Valid until the reader produces the next batch."""

    def __init__(self, reader, n, ptrs, raw_len, runs):
        self._reader, self._n, self._raw_len, self.runs = reader, n, raw_len, runs
        self.binned_for = None          # the CoverageHistogram whose bin size the "packed" column was written for (None: generic packed records)
        self.dev = {k: int(ptrs[i] or 0) for i, (k, _) in enumerate(_FIELDS)}
        self.dev["raw"] = int(ptrs[13] or 0)
        self._host = {}
        self._live = True

    def __len__(self):
        return self._n

    def __getattr__(self, name):
        if name.startswith("_") or name not in self.dev:
            raise AttributeError(name)
        if name not in self._host:
            if not self._live:
                raise RuntimeError("DeviceBatch used after the reader moved on to the next batch")
            if name == "raw":
                a = np.empty(self._raw_len, dtype=np.uint8)
            elif name == "packed":
                a = np.empty(self._n, dtype=np.uint64)
            else:
                a = np.empty(self._n, dtype=dict(_FIELDS)[name])
            ctx = self._reader.ctx
            _native.check(ctx.lib.tdt_copy_to_host(ctx.handle, _native.ptr(a), self.dev[name], a.nbytes))
            self._host[name] = a
        return self._host[name]

    def record(self, i):
        return RecordView(self, i)

    def release(self):
        """free the device buffers of a RETAINED batch (DeviceBamReader.retain); a no-op for ordinary batches"""
        h = getattr(self, "_retained", None)
        if h:
            self._reader.ctx.lib.tdt_ingest_release(h)
            self._retained = None
            self._live = False


class ScanCarry:
    """What `tiddit --sv`'s library-statistics pass hands to the signal pass of the same file: the reader (positioned behind the
    sampled batches), its batch iterator, the retained batches and the 50-bp histogram their coverage records were written for."""

    def __init__(self, path, reader, iterator, batches, hist):
        import os
        self.path, self.stamp = os.path.abspath(path), os.stat(path).st_mtime_ns
        self.reader, self.iterator, self.batches, self.hist = reader, iterator, batches, hist
        self.shard = getattr(reader, "shard", None)          # (rank, world) when the reader covers one rank's byte range of the file

    def drop(self):
        for b in self.batches:
            b.release()
        self.batches = []
        try:
            self.iterator.close()
        except Exception:
            pass
        self.reader.close()
        if self.hist is not None:
            self.hist.close()


_CARRY = None


def set_carry(carry):
    global _CARRY
    if _CARRY is not None:
        _CARRY.drop()
    _CARRY = carry


def take_carry(path, bin_size, shard=None):
    """the carry of `path` if this process left one for that bin size and that share of the file (the caller owns it then), else None"""
    global _CARRY
    import os
    c, _CARRY = _CARRY, None
    if c is None:
        return None
    try:
        ok = c.path == os.path.abspath(path) and c.stamp == os.stat(path).st_mtime_ns and c.hist.bin_size == int(bin_size) and \
            (None if c.shard is None else tuple(c.shard)) == (None if shard is None else tuple(shard))
    except OSError:
        ok = False
    if not ok:
        c.drop()
        return None
    return c


def preingest(path, shard, bin_size, stop, max_batches=12, chunk=448 << 20, ctx=None):
    """Start on this rank's share of the file before the scan's parameters are known: the N-rank `tiddit --sv` needs the library
    statistics (rank 0 samples them from the head of the file) before any signal predicate can run, but inflate, record decode and the
    coverage records depend on nothing — so the other ranks ingest their first batches meanwhile and keep them in HBM (retained, like
    the statistics pass's own).  Reads batches until ``stop()`` is true, the share is exhausted or `max_batches` are held (a batch
    holds ~1.8 GB of device memory at the default span); leaves a :class:`ScanCarry` for :func:`take_carry`.  -> batches held."""
    from . import tiddit_coverage
    reader = DeviceBamReader(path, ctx=ctx, chunk=chunk, shard=shard)
    hist = tiddit_coverage.CoverageHistogram([(n, l) for n, l in zip(reader.references, reader.lengths)], bin_size, ctx=reader.ctx)
    reader.bin_for(hist)
    reader.retain = True
    it = reader.batches()
    kept = []
    try:
        while len(kept) < max_batches and not stop():
            b = next(it, None)
            if b is None:
                break
            kept.append(b)
    except BaseException:
        for b in kept:
            b.release()
        it.close()
        reader.close()
        hist.close()
        raise
    reader.retain = False
    set_carry(ScanCarry(path, reader, it, kept, hist))
    return len(kept)


DEFAULT_RAMP_MB = 0              # first span of a reader in MB (0: every span is a full one), see DeviceBamReader._spans
_SPAN_POOLS = []                 # free list of hostutil.PinnedPool objects, each holding a reader's four span buffers
_SPAN_POOLS_LOCK = __import__("threading").Lock()


class DeviceBamReader:
    """BAM reader whose inflate, record finding and field decode run on the MI355X (``tdt_ingest_*``): the file's BGZF
    blocks are read into pinned host memory by a helper thread, pushed as they are, and come back as :class:`DeviceBatch`.
    Same ``header`` / ``references`` / ``lengths`` / ``batches()`` interface as :class:`BamReader`."""

    def __init__(self, path, ctx=None, chunk=448 << 20, shard=None, split_small=True):
        """split_small: a byte range shorter than four spans is read as four spans (see _spans); False = spans of `chunk` bytes whatever
        the file's size.  shard = (rank, world): read only the BGZF blocks that start in this rank's byte range of the file; the records
        that start in them are this shard's.  After ``batches()`` is exhausted, ``first_off`` / ``next_off`` hold the seam
        offsets that neighbouring shards must agree on (``dist.check_seams``)."""
        host = BamReader(path, batch_bytes=1 << 20)                 # the header is parsed on the host
        self.header, self.references, self.lengths, self.text = host.header, host.references, host.lengths, host.text
        self._skip = host.header_bytes
        host.close()
        self.path, self.chunk, self.split_small = path, chunk, split_small
        self.ctx = ctx or _native.default_context()
        self._f = open(path, "rb", buffering=0)
        h = ctypes.c_void_p()
        _native.check(self.ctx.lib.tdt_ingest_create(self.ctx.handle, len(self.references), ctypes.byref(h)))
        self._h = h
        self.host_chases = 0
        self._gen = None                 # the running batches() generator's state (ahead())
        self.spans_ahead = 0             # spans whose inflate was enqueued by ahead()
        self.reader_seconds = {"read": 0.0, "block scan": 0.0, "block table + copy issue": 0.0, "waited for the consumer": 0.0, "spans": 0,
                               "consumer waited for a span": 0.0}
        import os
        import threading
        self._stop = threading.Event()
        fsize = os.path.getsize(path)
        self.shard = shard
        self.retain = False              # True: every batch keeps its device buffers until DeviceBatch.release() (tdt_ingest_retain)
        self.collect_timing = os.environ.get("TIDDIT_INGEST_TIMING") == "1"     # per-batch stage times in self.timings (tdt_ingest_timing)
        self.timings = []
        self.first_off = self.next_off = None
        if shard is None:
            self._b_lo, self._b_hi, self._x_hi = 0, fsize, fsize
        else:
            r, w = shard
            self._b_lo = find_block_start(path, fsize * r // w)
            self._b_hi = fsize if r == w - 1 else find_block_start(path, fsize * (r + 1) // w)
            x = self._b_hi                                          # a few blocks past the shard complete its last record
            with open(path, "rb") as f:
                for _ in range(8):
                    if x >= fsize:
                        break
                    f.seek(x + 16)
                    x += struct.unpack("<H", f.read(2))[0] + 1
            self._x_hi = min(x, fsize)

    def bin_for(self, hist):
        """From the next batch on, the 8-byte coverage records the ingest kernel writes are BINNED records for `hist`'s bin size
        (CoverageHistogram.push_device_batch then takes the cheaper launch); None returns to the generic packed records.  -> whether
        the reader now writes binned records (bin sizes 1 and >= 1024 have none)."""
        on = ctypes.c_int(0)
        _native.check(self.ctx.lib.tdt_ingest_bin_for(self._h, hist.handle if hist is not None else None, ctypes.byref(on)))
        self._binned_for = hist if on.value else None
        return bool(on.value)

    def _spans(self):
        """(buffer, consumed) spans of whole BGZF blocks, read ahead by a helper thread into rotating pinned buffers"""
        import queue
        import threading
        lib = self.ctx.lib
        chunk = self.chunk
        b_lo, x_hi = self._b_lo, self._x_hi
        # a byte range shorter than four spans is read as four (not below 32 MB each): a file that fits ONE span used to go through read ->
        # copy -> inflate -> decode with nothing overlapping (bench.py's 259-MB file: 27.7 ms against 23.0 in 65-MB spans); short spans cost
        # nothing any more since their inflate kernels overlap on the reader's two streams
        if self.split_small:
            chunk = min(chunk, max((x_hi - b_lo + 3) // 4, 32 << 20))
        chunk = max(1 << 16, min(chunk, x_hi - b_lo))
        # pinned span buffers from a process-wide free list: handed back in Spans.close(), not whenever the collector gets to this
        # closure — a reader opened right after another one used to find the previous reader's pinned blocks still alive and paid
        # 5-55 ms for new ones (the spread of the bench's ingest line)
        with _SPAN_POOLS_LOCK:
            span_pool = _SPAN_POOLS.pop() if _SPAN_POOLS else None
        if span_pool is None:
            from .hostutil import PinnedPool
            span_pool = PinnedPool()
        # TWO helper threads (round 6): one READS the file into the next free pinned buffer, the other hops over the BGZF block headers
        # of what was read (tdt_bgzf_scan: a serial chase, ~4 ms per 448 MB), starts the span's PCIe copy and uploads its block table
        # (tdt_ingest_prefetch).  As one thread — read, then scan, then the next read — they took 13.7 ms per 448-MB span at 3 Gb, as
        # long as the span's inflate kernel: with the inflate running back to back the reader had become the other limit of the scan.
        # Five buffers: being read, being scanned, queued, and the consumer's two (the span it pushes and the one begun ahead).  The first
        # is pinned here, the others by the reading thread when it first needs them: pinning all of them up front was 0.2-0.3 s in
        # front of a process's first span, and the device had nothing to do meanwhile.
        NBUF, GAP = 5, 1 << 20            # GAP: data is read this far into a buffer; the partial block carried from the span before goes in front
        cap = chunk + GAP + (2 << 20)
        bufs = [span_pool.take("span0", cap, np.uint8)] + [None] * (NBUF - 1)
        q1 = queue.Queue(maxsize=1)       # read -> scan
        q = queue.Queue(maxsize=1)        # scan -> consumer
        free_q = queue.Queue()
        for i in range(NBUF):
            free_q.put(i)
        refs_lock = threading.Lock()
        refs = [0] * NBUF                 # spans of a buffer the consumer has not finished with
        closed = [False] * NBUF           # ... and whether the scanning thread is done with it
        stop = self._stop
        # TIDDIT_INGEST_RAMP=<MB>: the first span is that short and the following ones double up to the full span, so that the device does
        # not wait for 448 MB to be read and copied before its first kernel ("1" = 64; 0 = every span full).  Measured in rounds 3-5 and
        # again in round 6 (inflate kernels overlapping on two streams): no gain (profiles/r06_sv_ramp_240mb.txt)
        ramp = int(os.environ.get("TIDDIT_INGEST_RAMP", str(DEFAULT_RAMP_MB)) or 0)
        ramp = 64 if ramp == 1 else max(0, ramp)
        RS = self.reader_seconds                                   # where the helper threads' time goes, summed over the spans

        def put(qq, item):
            while not stop.is_set():
                try:
                    qq.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def get(qq):
            while not stop.is_set():
                try:
                    return qq.get(timeout=0.05)
                except queue.Empty:
                    pass
            return None

        def release_buffer(idx):
            """the consumer is done with one span of buffer idx"""
            with refs_lock:
                refs[idx] -= 1
                done = refs[idx] == 0 and closed[idx]
                if done:
                    closed[idx] = False
            if done:
                free_q.put(idx)

        def read_loop():
            from concurrent.futures import ThreadPoolExecutor
            fd, fsize, fo = self._f.fileno(), x_hi, b_lo           # this reader's byte range of the file
            pool = ThreadPoolExecutor(int(os.environ.get("TIDDIT_READ_THREADS", "16")))       # (8 threads: 9.2 GB/s of BGZF from the page cache, 16: 10.5)
            try:
                lib.tdt_ctx_bind_thread(self.ctx.handle)           # (this thread pins buffers: on the context's device)
                k = 0
                while fo < fsize:
                    t_w = time.perf_counter()
                    idx = get(free_q)
                    RS["reader waited for a buffer"] = RS.get("reader waited for a buffer", 0.0) + time.perf_counter() - t_w
                    if idx is None:
                        return
                    if bufs[idx] is None:
                        bufs[idx] = span_pool.take("span%d" % idx, cap, np.uint8)
                    buf = bufs[idx]
                    ck = min(chunk, max(1 << 16, (ramp << 20) << min(k, 8))) if ramp else chunk
                    want = fsize - fo if fsize - fo <= ck + (1 << 20) else ck         # the tail rides along
                    piece = 8 << 20
                    mv = memoryview(buf)

                    def rd(o):                                   # parallel positional reads into the pinned buffer
                        n, end = 0, min(want, o + piece)
                        while o + n < end:
                            g = os.preadv(fd, [mv[GAP + o + n:GAP + end]], fo + o + n)
                            if g <= 0:
                                raise ValueError("short read")
                            n += g
                        return n
                    t_rd = time.perf_counter()
                    got = sum(pool.map(rd, range(0, want, piece)))
                    read_ms = 1e3 * (time.perf_counter() - t_rd)
                    RS["read"] += read_ms * 1e-3
                    if not put(q1, (idx, got, fo, read_ms, fo + got >= fsize)):
                        return
                    fo += got
                    k += 1
                put(q1, None)
            except BaseException as e:
                put(q1, e)
            finally:
                pool.shutdown(wait=False)

        def scan_loop():
            try:
                carry = np.zeros(0, dtype=np.uint8)
                while True:
                    item = get(q1)
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    idx, got, fo0, read_ms, eof = item
                    buf = bufs[idx]
                    if len(carry) > GAP:
                        raise ValueError("BGZF block larger than the read window")
                    pos = GAP - len(carry)
                    buf[pos:GAP] = carry
                    end = GAP + got
                    emitted = 0
                    # (one scan covers the buffer unless its blocks inflate to more than a push takes — 3 GiB: highly compressible data —,
                    # then the rest of the same buffer goes out as further spans)
                    while pos < end:
                        t_sc = time.perf_counter()
                        nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
                        view = buf[pos:end]
                        _native.check(lib.tdt_bgzf_scan(_native.ptr(view), end - pos, 3 << 30, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
                        t_pf = time.perf_counter()
                        RS["block scan"] += t_pf - t_sc
                        if nb.value == 0:
                            break
                        # the PCIe copy of this span starts now, on the copy stream, behind whatever the device is doing for the span before it
                        # (the push of exactly this (pointer, length) then finds it on the device)
                        _native.check(lib.tdt_ingest_prefetch(self._h, _native.ptr(view), consumed.value))
                        t_put = time.perf_counter()
                        RS["block table + copy issue"] += t_put - t_pf
                        with refs_lock:
                            refs[idx] += 1
                        ok_put = put(q, (view, consumed.value, fo0 - (GAP - pos), read_ms if not emitted else 0.0, idx))
                        RS["waited for the consumer"] += time.perf_counter() - t_put
                        RS["spans"] += 1
                        if not ok_put:
                            return
                        emitted += 1
                        pos += consumed.value
                    carry = buf[pos:end].copy()
                    with refs_lock:
                        closed[idx] = True
                        idle = refs[idx] == 0
                        if idle:
                            closed[idx] = False
                    if idle:
                        free_q.put(idx)
                    if len(carry) and eof:
                        raise ValueError("truncated BGZF block at end of file")
                    if len(carry) > 1 << 16 and not emitted:
                        raise ValueError("BGZF block larger than the read window")
                put(q, None)
            except BaseException as e:
                put(q, e)

        th = threading.Thread(target=scan_loop, daemon=True)
        th_read = threading.Thread(target=read_loop, daemon=True)
        th_read.start()
        th.start()
        self._span_thread = th                                    # close() joins both before the ingest handle goes (the scanning thread prefetches through it)
        self._read_thread = th_read

        class Spans:
            """blocking ``next()`` and non-blocking ``poll()`` over the reader thread's queue; None = end of range"""
            done = False
            returned = False

            release = staticmethod(release_buffer)

            def _take(self, item):
                if item is None:
                    self.done = True
                    stop.set()
                    th.join()
                    th_read.join()
                    return None
                if isinstance(item, BaseException):
                    self.done = True
                    stop.set()
                    raise item
                return item

            def next(self):
                return None if self.done else self._take(q.get())

            def poll(self):
                if self.done:
                    return None
                try:
                    return self._take(q.get_nowait())
                except queue.Empty:
                    return False                                  # nothing read yet

            def close(self):
                stop.set()                                        # a consumer that stops early releases the helper threads
                th.join()
                th_read.join()
                if not self.returned:
                    self.returned = True
                    with _SPAN_POOLS_LOCK:
                        _SPAN_POOLS.append(span_pool)

        return Spans()

    def batches(self):
        """yield :class:`DeviceBatch` objects, one per span of BGZF blocks pushed through the device ingest"""
        if self.shard is not None and self._b_lo >= self._b_hi:      # an empty shard: the seam passes straight through
            return
        spans = self._spans()
        try:
            yield from self._batches(spans, self.ctx.lib, self.ctx, ctypes.c_size_t(-1).value)
        finally:
            spans.close()

    def _begin(self, st, item):
        """first half of `item`'s push (tdt_ingest_push_ahead): its inflate + CRC kernels go to the reader's inflate streams now"""
        _native.check(self.ctx.lib.tdt_ingest_push_ahead(self._h, _native.ptr(item[0]), item[1]))
        self.spans_ahead += 1

    def ahead(self):
        """Start the NEXT span's inflate if the reader thread has the span and it was not started yet.  ``batches()`` does this by itself
        before it parses the current span (the next span's inflate runs on the reader's own streams, into its own output buffer, and
        depends on nothing the current batch's consumers do: ``tdt_ingest_push_ahead``); a consumer calls it once more after launching its
        kernels in case the reader thread had not delivered the span then.  The current batch — field arrays AND raw bytes — stays valid
        until the generator is advanced.  -> whether a span was started by this call."""
        st = self._gen
        if st is None or st["ahead"] or os.environ.get("TIDDIT_INGEST_AHEAD", "1") == "0":
            return False
        if st["pending"] is False:
            st["pending"] = st["spans"].poll()
        item = st["pending"]
        if item is False or item is None:
            return False
        self._begin(st, item)
        st["ahead"] = True
        return True

    def _batches(self, spans, lib, ctx, nothing):
        prev, first = None, True
        # pending — False: span k+1 not looked at yet; None: end of the range; else the span itself.  (On the reader: ahead() looks at it.)
        st = self._gen = {"pending": False, "spans": spans, "ahead": False}
        try:
            yield from self._batches_loop(st, spans, lib, ctx, nothing, prev, first)
        finally:
            self._gen = None

    def _batches_loop(self, st, spans, lib, ctx, nothing, prev, first):
        while True:
            t_wait = time.perf_counter()
            cur = spans.next() if st["pending"] is False else st["pending"]
            wait_ms = 1e3 * (time.perf_counter() - t_wait)
            self.reader_seconds["consumer waited for a span"] += wait_ms * 1e-3
            if cur is None:
                break
            buf, consumed, abs0, read_ms, buf_idx = cur
            begun, st["ahead"] = st["ahead"], False                 # (a span started by ahead() is this one: its push finishes below)
            st["pending"] = spans.poll()                            # span k+1 already read?  (its PCIe copy was started by the reader thread)
            if os.environ.get("TIDDIT_INGEST_AHEAD", "1") != "0":
                # span k+1 goes to the inflate streams BEFORE span k is parsed: the chip never leaves the inflate kernel while the record
                # search, the field decode and the consumer's kernels of batch k run beside it (pushes must follow the order of the begins)
                if not begun:
                    self._begin(st, cur)
                    self.spans_ahead -= 1                           # (not ahead of anything: counted are spans begun before their turn)
                self.ahead()
            if prev is not None and not getattr(prev, "_retained", None):
                prev._live = False
            n = ctypes.c_size_t(0)
            skip = (self._skip if self._b_lo == 0 else nothing) if first else 0
            final = self.shard is not None and abs0 + consumed >= self._x_hi
            if self.shard is not None and not final and abs0 + consumed > self._b_hi:
                raise RuntimeError("shard tail split across reads")
            if final:
                own_c = max(0, min(consumed, self._b_hi - abs0))
                nb_, c_, own = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
                _native.check(lib.tdt_bgzf_scan(_native.ptr(buf), own_c, 3 << 30, ctypes.byref(nb_), ctypes.byref(c_), ctypes.byref(own)))
                if c_.value != own_c:
                    raise RuntimeError("shard boundary is not a block boundary")
                fo, no = ctypes.c_size_t(0), ctypes.c_size_t(0)
                _native.check(lib.tdt_ingest_push_bounded(self._h, _native.ptr(buf), consumed, skip, own.value, ctypes.byref(n),
                                                          ctypes.byref(fo), ctypes.byref(no)))
                if first:
                    self.first_off = fo.value
                self.next_off = no.value
            elif first and self.shard is not None:
                fo = ctypes.c_size_t(0)
                _native.check(lib.tdt_ingest_push_bounded(self._h, _native.ptr(buf), consumed, skip, nothing, ctypes.byref(n), ctypes.byref(fo), None))
                self.first_off = fo.value
            else:
                _native.check(lib.tdt_ingest_push(self._h, _native.ptr(buf), consumed, skip, ctypes.byref(n)))
            first = False
            spans.release(buf_idx)                                 # (the push has waited for the span's inflate: its host bytes are not needed again)
            if self.collect_timing:
                tm = (ctypes.c_double * 8)()
                _native.check(lib.tdt_ingest_timing(self._h, tm))
                self.timings.append({"records": int(n.value), "bgzf_MB": consumed / 1e6, "read_ms": read_ms, "wait_for_reader_ms": wait_ms,
                                     "block_table_ms": tm[0], "h2d_ms": tm[1], "inflate_crc_ms": tm[2], "find_records_ms": tm[3], "chain_check_ms": tm[4],
                                     "decode_ms": tm[5], "h2d_prefetched": bool(tm[6]), "push_wall_ms": tm[7]})
            if not n.value:
                continue
            ptrs = (ctypes.c_void_p * 14)()
            raw_len = ctypes.c_size_t(0)
            _native.check(lib.tdt_ingest_arrays(self._h, ptrs, ctypes.byref(raw_len)))
            edges = np.empty(8192, dtype=np.uint32)
            ne = ctypes.c_size_t(0)
            _native.check(lib.tdt_ingest_edges(self._h, _native.ptr(edges), 8192, ctypes.byref(ne)))
            b = DeviceBatch(self, n.value, ptrs, raw_len.value, None)
            pk = ctypes.c_void_p()
            _native.check(lib.tdt_ingest_packed(self._h, ctypes.byref(pk)))
            b.dev["packed"] = int(pk.value or 0)         # 8-byte coverage records (csrc/tdt_common.h: cov_pack_record / cov_bin_record)
            b.binned_for = getattr(self, "_binned_for", None)
            if ne.value == ctypes.c_size_t(-1).value:                   # not coordinate sorted: runs from the tid column
                tid = b.tid
                lo = np.concatenate([[0], np.flatnonzero(np.diff(tid)) + 1])
                tids = tid[lo]
            else:
                lo = edges[:ne.value].astype(np.int64)
                tids = np.empty(max(1, ne.value), dtype=np.int32)    # the contig of every run came back with the edges
                _native.check(lib.tdt_ingest_edge_tids(self._h, _native.ptr(tids), ne.value))
                tids = tids[:ne.value]
            hi = np.concatenate([lo[1:], [n.value]])
            b.runs = [(int(t), int(l), int(h)) for t, l, h in zip(tids, lo, hi)]
            if self.retain:
                rh = ctypes.c_void_p()
                _native.check(lib.tdt_ingest_retain(self._h, ctypes.byref(rh)))
                b._retained = rh
            prev = st["batch"] = b
            yield b
        c, hc = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _native.check(lib.tdt_ingest_carry(self._h, ctypes.byref(c), ctypes.byref(hc)))
        self.host_chases = hc.value
        if c.value:
            raise ValueError("truncated BAM record at end of file")
        if self.shard is not None and self.shard[0] == self.shard[1] - 1 and self.next_off not in (None, 0):
            raise ValueError("truncated BAM record at end of file")

    def close(self):
        """safe whatever the lifetime of a ``batches()`` generator: the reader thread (which calls tdt_ingest_prefetch on the handle and
        reads the file) is stopped and joined before the handle is destroyed and the file closed"""
        self._stop.set()
        for attr in ("_span_thread", "_read_thread"):
            th = getattr(self, attr, None)
            if th is not None and th.is_alive() and th is not __import__("threading").current_thread():
                th.join()
            setattr(self, attr, None)
        if self._h:
            self.ctx.lib.tdt_ingest_destroy(self._h)
            self._h = None
        self._f.close()


def open_bam(path, ctx=None):
    """The reader the pipeline uses: inflate + decode on the device (:class:`DeviceBamReader`); ``TIDDIT_HOST_INGEST=1``
    selects the host-thread reader (:class:`BamReader`), whose batches are numpy-backed."""
    import os
    if os.environ.get("TIDDIT_HOST_INGEST") == "1":
        return BamReader(path)
    return DeviceBamReader(path, ctx=ctx, chunk=int(os.environ.get("TIDDIT_INGEST_CHUNK", str(448 << 20))))


# ------------------------------------------------------------------------------------ writer
def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def parse_cigar(cigar):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((CIGAR_OPS.index(ch), int(num)))
            num = ""
    return out


def encode_record(qname, flag, tid, pos, mapq, cigar, mate_tid, mate_pos, tlen, seq="", tags=(), qual=None):
    """One BAM alignment record (block_size included) as bytes.  pos 0-based; cigar = string or list of (op, len);
    tags = iterable of (tag, type, value); qual = bytes of l_seq phred values or None (0xff: absent)"""
    cig = parse_cigar(cigar) if isinstance(cigar, str) else list(cigar)
    rlen = sum(l for op, l in cig if op in (0, 2, 3, 7, 8)) or 1
    name = qname.encode() + b"\x00"
    lseq = len(seq)
    codes = [_SEQ_CODES.index(c) if c in _SEQ_CODES else 15 for c in seq.upper()]
    if lseq & 1:
        codes.append(0)
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    aux = b""
    for tag, typ, val in tags:
        if typ == "Z":
            aux += tag.encode() + b"Z" + val.encode() + b"\x00"
        elif typ == "i":
            aux += tag.encode() + b"i" + struct.pack("<i", val)
        elif typ == "A":
            aux += tag.encode() + b"A" + val.encode()
        elif typ in "cCsSIf":
            aux += tag.encode() + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "I": "I", "f": "f"}[typ], val)
        elif typ == "H":
            aux += tag.encode() + b"H" + val.encode() + b"\x00"
        elif typ.startswith("B"):                                  # ("XB", "BS", [1, 2, 3]): array of subtype S
            sub = typ[1]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            aux += tag.encode() + b"B" + sub.encode() + struct.pack("<I", len(val)) + struct.pack("<%d%s" % (len(val), fmt), *val)
        else:
            raise ValueError("unsupported tag type " + typ)
    body = (struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, _reg2bin(pos, pos + rlen), len(cig), flag, lseq, mate_tid,
                        mate_pos, tlen) + name + b"".join(struct.pack("<I", (l << 4) | op) for op, l in cig) + packed +
            (b"\xff" * lseq if qual is None else bytes(qual)) + aux)
    return struct.pack("<i", len(body)) + body


class BamWriter:
    """Write a BAM from Python values (test / synthetic-config tooling)."""

    def __init__(self, path, references, text=None, level=1, align_records=False):
        """align_records: start a new BGZF block rather than split a record across two (what htslib's writer does);
        the default packs blocks to 0xff00 bytes regardless of record boundaries (htsjdk style)."""
        self._f = open(path, "wb")
        self._buf = bytearray()
        self.level = level
        self.align_records = align_records
        self.references = [r[0] for r in references]
        if text is None:
            text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in references)
        t = text.encode()
        self._buf += b"BAM\x01" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(references))
        for name, ln in references:
            nb = name.encode() + b"\x00"
            self._buf += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)

    def write(self, qname, flag, tid, pos, mapq, cigar, mate_tid, mate_pos, tlen, seq="", tags=()):
        """pos 0-based; cigar = string or list of (op, len); tags = iterable of (tag, type, value) with type Z/i/A"""
        body = encode_record(qname, flag, tid, pos, mapq, cigar, mate_tid, mate_pos, tlen, seq, tags)[4:]
        if self.align_records and self._buf and len(self._buf) + 4 + len(body) > 0xff00:
            self._f.write(_bgzf_block(bytes(self._buf), self.level))
            self._buf = bytearray()
        self._buf += struct.pack("<i", len(body)) + body
        while len(self._buf) >= 0xff00:
            self._f.write(_bgzf_block(bytes(self._buf[:0xff00]), self.level))
            del self._buf[:0xff00]

    def close(self):
        while self._buf:
            self._f.write(_bgzf_block(bytes(self._buf[:0xff00]), self.level))
            del self._buf[:0xff00]
        self._f.write(_BGZF_EOF)
        self._f.close()
