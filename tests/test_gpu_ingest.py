"""GPU parity of the device ingest: BGZF inflate (one wavefront per block) against zlib, record finding + field decode
against the host decoder (tdt_bam_decode, itself checked against the independent parser in test_bamio_cpu.py), and the
pipeline (tiddit --cov / signal scan) with device ingest against the host-thread ingest."""
import os
import struct
import zlib

import numpy as np
import pytest

from tiddit_amd import _native, bamio, synth_bam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return _native.default_context()


def _bgzf(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, block=0xff00):
    out = b""
    for o in range(0, max(1, len(data)), block):
        d = data[o:o + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        comp = c.compress(d) + c.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return out + bamio._BGZF_EOF


def _inflate_hbm(ctx, comp, n):
    comp = np.frombuffer(comp, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)
    _native.check(ctx.lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), len(comp), _native.ptr(out), n, 0))
    return out.tobytes()


def _cases():
    rng = np.random.default_rng(1)
    return {
        "empty": b"", "one": b"A", "three": b"abc", "zeros": bytes(200000),
        "text": b"the quick brown fox jumps over the lazy dog. " * 5000,
        "random": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(),
        "dna": np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, 300000)].tobytes(),
        "skew": np.minimum(255, rng.geometric(0.05, 300000)).astype(np.uint8).tobytes(),
        "runs": b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 3000), rng.integers(1, 400, 3000))),
        "period": (b"abcdefg" * 30000) + (b"xy" * 20000) + (b"0123456789ABCDEF" * 9000),
    }


@pytest.mark.parametrize("kernel", ["lanes", "sequential"])
@pytest.mark.parametrize("name", list(_cases()))
def test_device_inflate_matches_zlib(ctx, name, kernel, monkeypatch):
    """stored / fixed / dynamic DEFLATE blocks, long codes, overlapping matches, 0..64 KiB blocks; CRC32 checked on the device.
    Both kernels: the lane-parallel default (tdt_inflate2.hip) and the one-symbol-at-a-time one (TIDDIT_INFLATE_SEQ=1)."""
    if kernel == "sequential":
        monkeypatch.setenv("TIDDIT_INFLATE_SEQ", "1")
    data = _cases()[name]
    for level in (0, 1, 6, 9):
        assert _inflate_hbm(ctx, _bgzf(data, level), len(data)) == data, (name, level)
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        assert _inflate_hbm(ctx, _bgzf(data, 6, strategy), len(data)) == data, (name, strategy)
    assert _inflate_hbm(ctx, _bgzf(data, 6, block=777), len(data)) == data          # many small blocks


def test_device_inflate_rejects_damage(ctx):
    data = _cases()["text"]
    comp = bytearray(_bgzf(data, 6))
    good = bytes(comp)
    comp[len(comp) // 3] ^= 0x40                                  # payload bit flip
    with pytest.raises(_native.TdtError):
        _inflate_hbm(ctx, bytes(comp), len(data))
    bad_crc = bytearray(good)
    first_bsize = struct.unpack_from("<H", good, 16)[0] + 1
    bad_crc[first_bsize - 8] ^= 1                                 # CRC32 field of the first block
    with pytest.raises(_native.TdtError) as e:
        _inflate_hbm(ctx, bytes(bad_crc), len(data))
    assert "CRC32" in str(e.value)
    with pytest.raises(_native.TdtError):                         # caller's size disagrees with ISIZE
        _inflate_hbm(ctx, good, len(data) - 1)
    with pytest.raises(_native.TdtError):                         # not a whole number of blocks
        _inflate_hbm(ctx, good[:-40], len(data))
    assert _inflate_hbm(ctx, good, len(data)) == data             # the context still works afterwards


FIELDS = [k for k, _ in bamio._FIELDS]


def _host_records(path):
    r = bamio.BamReader(path)
    cols = {k: [] for k in FIELDS}
    sa, names = [], []
    for b in r.batches():
        for k in FIELDS:
            cols[k].append(getattr(b, k))
        sa.extend(b.record(int(i)).get_tag_sa() for i in np.flatnonzero(b.sa_off >= 0))
        names.extend(b.record(int(i)).query_name for i in range(0, len(b), 997))
    r.close()
    return {k: np.concatenate(v) for k, v in cols.items()}, sa, names


def _device_records(path, ctx, chunk):
    r = bamio.DeviceBamReader(path, ctx=ctx, chunk=chunk)
    cols = {k: [] for k in FIELDS}
    sa, runs_ok, nb = [], True, 0
    for b in r.batches():
        nb += 1
        for k in FIELDS:
            cols[k].append(getattr(b, k))
        sa.extend(b.record(int(i)).get_tag_sa() for i in np.flatnonzero(b.sa_off >= 0))
        lo = np.concatenate([[0], np.flatnonzero(np.diff(b.tid)) + 1])
        want = [(int(b.tid[l]), int(l), int(h)) for l, h in zip(lo, np.concatenate([lo[1:], [len(b)]]))]
        runs_ok &= want == b.runs
        i = len(b) // 2                                            # rec_off points at the record inside the batch's raw bytes
        assert b.raw[int(b.rec_off[i]) + 4:int(b.rec_off[i]) + 8].view(np.int32)[0] == b.tid[i]
    hc = r.host_chases
    r.close()
    return {k: np.concatenate(v) for k, v in cols.items()}, sa, runs_ok, nb, hc


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("ingest")
    sv = str(d / "sv.bam")
    synth_bam.write_synthetic_bam(sv, [("chr1", 300000), ("chr2", 200000), ("chrM", 3000), ("tiny", 500)], depth=8, seed=5)
    bulk = str(d / "bulk.bam")
    synth_bam.write_bulk_bam(bulk, [("chr1", 1_500_000), ("chr2", 1_000_000)], depth=30, threads=8)
    return sv, bulk


@pytest.mark.parametrize("which,chunk", [(0, 1 << 28), (0, 300_000), (1, 1 << 28), (1, 2_000_000), (1, 200_000)])
def test_device_ingest_matches_host_decode(ctx, bams, which, chunk):
    """all thirteen field arrays, the SA strings and the contig runs; records and BGZF blocks straddle every push boundary"""
    path = bams[which]
    want, sa_w, _ = _host_records(path)
    got, sa_g, runs_ok, nb, hc = _device_records(path, ctx, chunk)
    for k in FIELDS:
        if k in ("rec_off", "sa_off"):                            # batch-relative; checked through the strings / raw bytes
            continue
        assert np.array_equal(want[k], got[k]), k
    assert np.array_equal(want["sa_off"] >= 0, got["sa_off"] >= 0) and sa_w == sa_g and runs_ok
    assert hc == 0                                                # every record chain was confirmed from the device guesses
    if chunk < 1 << 20:
        assert nb > 3


def test_device_ingest_host_chase_fallback(ctx, bams, monkeypatch):
    """the serial host chase that backs up the per-segment guesses gives the same records"""
    want, sa_w, _ = _host_records(bams[0])
    monkeypatch.setenv("TIDDIT_INGEST_HOST_CHASE", "1")
    got, sa_g, runs_ok, nb, hc = _device_records(bams[0], ctx, 300_000)
    assert hc == nb and nb > 3 and runs_ok and sa_w == sa_g
    for k in ("tid", "pos", "end", "mapq", "flag", "mate_tid", "mate_pos", "tlen", "l_seq", "cigar_first", "cigar_last"):
        assert np.array_equal(want[k], got[k]), k


def test_device_ingest_truncated_file(ctx, bams, tmp_path):
    raw = open(bams[0], "rb").read()
    blocks, o = [], 0
    while o < len(raw):
        bs = struct.unpack_from("<H", raw, o + 16)[0] + 1
        blocks.append(raw[o:o + bs])
        o += bs
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(b"".join(blocks[:len(blocks) // 2]))     # ends on a block boundary, in the middle of a record
    with pytest.raises(ValueError):
        for _ in bamio.DeviceBamReader(cut, ctx=ctx).batches():
            pass


def test_cli_cov_device_ingest_equals_host_ingest(bams, tmp_path, monkeypatch):
    """tiddit --cov: BGZF inflate + decode + histogram in HBM vs host-thread inflate/decode + device histogram: same bytes"""
    from tiddit_amd import __main__ as cli
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TIDDIT_HOST_INGEST", mode)
        for z, wig in ((500, False), (50, True)):
            o = str(tmp_path / ("cov_%s_%d" % (mode, z)))
            cli.run_cov(cli._cov_parser().parse_args(["--cov", "--bam", bams[1], "-o", o, "-z", str(z)] + (["-w"] if wig else [])))
            outs[(mode, z)] = open(o + (".wig" if wig else ".bed")).read()
    assert outs[("1", 500)] == outs[("0", 500)] and outs[("1", 50)] == outs[("0", 50)] and len(outs[("0", 500)]) > 1000


def test_signal_scan_device_ingest_equals_host_ingest(bams, monkeypatch):
    """tiddit_signal.scan_signals (coverage, discordant pairs, splits, clips) is the same through either reader"""
    from tiddit_amd import tiddit_signal
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TIDDIT_HOST_INGEST", mode)
        res[mode] = tiddit_signal.scan_signals(bams[0], 5, 600, 10000, 30, 20)
    h, d = res["1"], res["0"]
    text = lambda clips: {c: b"".join(tiddit_signal._clip_bytes(e) for e in v) for c, v in clips.items()}     # (entries are only ever written joined)
    assert h[1] == d[1] and h[3] == d[3] and h[4] == d[4] and text(h[5]) == text(d[5])
    assert sum(len(v) for v in d[3].values()) > 0 and sum(len(v) for v in d[4].values()) > 0 and sum(len(v) for v in text(d[5]).values()) > 0
    for c in h[2]:
        assert np.array_equal(h[2][c], d[2][c])


@pytest.mark.parametrize("which,world,chunk", [(0, 2, 1 << 28), (0, 5, 200_000), (1, 3, 1 << 28), (1, 8, 1_000_000), (0, 64, 1 << 28)])
def test_sharded_device_ingest_is_the_sequential_decode(ctx, bams, which, world, chunk):
    """one BAM read as `world` byte-range shards (mid-file starts found by the record guess): the shards' records, in rank
    order, are exactly the unsharded decode, and every seam agrees (what dist.check_seams enforces across ranks)"""
    path = bams[which]
    want, _, _ = _host_records(path)
    cols = {k: [] for k in FIELDS}
    seams = []
    for r in range(world):
        rd = bamio.DeviceBamReader(path, ctx=ctx, chunk=chunk, shard=(r, world))
        for b in rd.batches():
            for k in FIELDS:
                cols[k].append(getattr(b, k))
        seams.append((rd.first_off, rd.next_off))
        assert rd.host_chases == 0
        rd.close()
    for k in FIELDS:
        if k in ("rec_off", "sa_off"):
            continue
        assert np.array_equal(want[k], np.concatenate(cols[k])), k
    live = [s for s in seams if s[0] is not None]
    assert len(live) >= min(world, 2) and live[-1][1] == 0
    for a, b in zip(live, live[1:]):
        assert a[1] == b[0]


def test_coverage_sharded_single_rank_group_equals_cli(bams, tmp_path, monkeypatch):
    """dist.coverage_sharded (sharded ingest + seam check + all-reduce) in a 1-rank process group vs the plain CLI path"""
    import socket
    import torch.distributed as dist
    from tiddit_amd import __main__ as cli
    from tiddit_amd import dist as tdist
    from tiddit_amd import tiddit_coverage
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        header, cov, n = tdist.coverage_sharded(bams[1], 500, 20)
    finally:
        dist.destroy_process_group()
    o = str(tmp_path / "a")
    tiddit_coverage.print_coverage(cov, header, 500, "bed", o + ".bed")
    cli.run_cov(cli._cov_parser().parse_args(["--cov", "--bam", bams[1], "-o", str(tmp_path / "b"), "-z", "500"]))
    assert open(o + ".bed").read() == open(str(tmp_path / "b.bed")).read() and n > 100000


def _write_odd_bam(path, kind, seed=11):
    """BAMs that stress the record finder: 'long' = reads far larger than a BGZF block / a 16 KiB segment between short
    ones, 'aligned' = htslib-style blocks that never split a record, 'unsorted' = contig ids change on almost every record"""
    rng = np.random.default_rng(seed)
    refs = [("c%d" % i, 2_000_000) for i in range(6)]
    w = bamio.BamWriter(path, refs, level=1, align_records=(kind == "aligned"))
    n = 0
    recs = []
    for i in range(3000):
        tid = int(rng.integers(0, len(refs))) if kind == "unsorted" else min(len(refs) - 1, i // 500)
        pos = int(rng.integers(0, 1_000_000)) if kind == "unsorted" else (i % 500) * 1500
        if kind == "long" and i % 40 == 7:
            ln = int(rng.integers(20_000, 150_000))
            seq = "".join(rng.choice(list("ACGT"), ln))
            cig = "%dS%dM%dI%dM" % (5, ln // 2, 3, ln - 5 - 3 - ln // 2)
        else:
            ln = int(rng.integers(30, 152))
            seq = "".join(rng.choice(list("ACGT"), ln))
            cig = "%dM" % ln
        tags = (("SA", "Z", "c1,%d,+,50M50S,30,0;" % (pos + 7)),) if i % 97 == 0 else ()
        if i % 5 == 0:
            tags = tags + (("NM", "i", int(i % 7)), ("RG", "Z", "grp"))
        recs.append(dict(qname="r%d/%s" % (i, "x" * int(rng.integers(0, 40))), flag=int(rng.choice([99, 147, 83, 163, 1024 + 99, 4])), tid=tid, pos=pos,
                         mapq=int(rng.integers(0, 61)), cigar=cig, mate_tid=tid, mate_pos=pos + 200, tlen=300, seq=seq, tags=tags))
    for r in recs:
        w.write(**r)
        n += 1
    w.close()
    return n


@pytest.mark.parametrize("kind,chunk", [("long", 1 << 28), ("long", 150_000), ("aligned", 1 << 28), ("aligned", 90_000), ("unsorted", 1 << 28)])
def test_device_ingest_odd_files(ctx, tmp_path, kind, chunk):
    path = str(tmp_path / (kind + ".bam"))
    n = _write_odd_bam(path, kind)
    want, sa_w, _ = _host_records(path)
    got, sa_g, runs_ok, nb, hc = _device_records(path, ctx, chunk)
    assert len(want["tid"]) == n == len(got["tid"])
    for k in FIELDS:
        if k not in ("rec_off", "sa_off"):
            assert np.array_equal(want[k], got[k]), k
    assert sa_w == sa_g and runs_ok and hc == 0


def test_cli_cov_unsorted_bam_device_equals_host(tmp_path, monkeypatch):
    """contig id changes on nearly every record: the device reader derives the runs from the tid column"""
    from tiddit_amd import __main__ as cli
    path = str(tmp_path / "u.bam")
    _write_odd_bam(path, "unsorted")
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TIDDIT_HOST_INGEST", mode)
        o = str(tmp_path / ("cov" + mode))
        cli.run_cov(cli._cov_parser().parse_args(["--cov", "--bam", path, "-o", o, "-z", "1000", "-q", "10"]))
        outs[mode] = open(o + ".bed").read()
    assert outs["0"] == outs["1"] and len(outs["0"]) > 1000


@pytest.mark.parametrize("kernel", ["lanes", "sequential"])
def test_device_inflate_damage_fuzz(ctx, kernel, monkeypatch):
    """random bit flips, smashed bytes, zeroed tails and 0xff runs inside BGZF payloads: every damaged stream is rejected
    (decode error or CRC32), nothing hangs, and the context still works afterwards (tools/fuzz_inflate_gpu.py runs more)"""
    if kernel == "sequential":
        monkeypatch.setenv("TIDDIT_INFLATE_SEQ", "1")
    rng = np.random.default_rng(99)
    cases = _cases()
    srcs = [cases["skew"], cases["text"], cases["dna"]]
    rejected = 0
    for it in range(60):
        data = srcs[it % 3]
        comp = bytearray(_bgzf(data, (1, 6, 9)[it % 3]))
        spans, o = [], 0
        while o < len(comp) - 28:
            bs = struct.unpack_from("<H", comp, o + 16)[0] + 1
            spans.append((o + 18, bs - 26))
            o += bs
        for _ in range(1 + it % 4):
            so, n = spans[int(rng.integers(0, len(spans)))]
            p = so + int(rng.integers(0, n))
            if it % 4 == 0:
                comp[p] ^= 1 << int(rng.integers(0, 8))
            elif it % 4 == 1:
                comp[p:p + 8] = bytes(rng.integers(0, 256, len(comp[p:p + 8]), dtype=np.uint8))
            elif it % 4 == 2:
                comp[p:so + n] = bytes(so + n - p)
            else:
                comp[p:min(p + 64, so + n)] = b"\xff" * (min(p + 64, so + n) - p)
        try:
            assert _inflate_hbm(ctx, bytes(comp), len(data)) == data      # a harmless flip must still give the right bytes
        except _native.TdtError:
            rejected += 1
    assert rejected >= 55
    assert _inflate_hbm(ctx, _bgzf(srcs[1], 6), len(srcs[1])) == srcs[1]


def test_device_ingest_rejects_damaged_records(ctx, tmp_path):
    """valid BGZF around a damaged BAM record: the device reader fails like the host decoder does (no crash, no garbage)"""
    path = str(tmp_path / "ok.bam")
    synth_bam.write_synthetic_bam(path, [("chr1", 120000), ("chr2", 50000)], depth=6, seed=8)
    raw = b"".join(bamio.bgzf_blocks(open(path, "rb")))
    hdr = bamio.BamReader(path)
    skip = hdr.header_bytes
    hdr.close()
    offs, o = [], skip
    while o + 4 <= len(raw):
        offs.append(o)
        o += 4 + struct.unpack_from("<I", raw, o)[0]
    for what in ("block_size", "l_seq", "n_cigar"):
        dmg = bytearray(raw)
        r = offs[len(offs) // 2]
        if what == "block_size":
            struct.pack_into("<I", dmg, r, 7)
        elif what == "l_seq":
            struct.pack_into("<i", dmg, r + 4 + 16, -5)
        else:
            struct.pack_into("<H", dmg, r + 4 + 12, 60000)
        bad = str(tmp_path / (what + ".bam"))
        with open(bad, "wb") as f:
            for k in range(0, len(dmg), 0xff00):
                f.write(bamio._bgzf_block(bytes(dmg[k:k + 0xff00]), 1))
            f.write(bamio._BGZF_EOF)
        for make in (lambda: bamio.BamReader(bad), lambda: bamio.DeviceBamReader(bad, ctx=ctx), lambda: bamio.DeviceBamReader(bad, ctx=ctx, chunk=100_000)):
            with pytest.raises((_native.TdtError, ValueError)):
                rd = make()
                for _ in rd.batches():
                    pass
    rd = bamio.DeviceBamReader(path, ctx=ctx)                         # the context is still healthy
    assert sum(len(b) for b in rd.batches()) == len(offs)
    rd.close()


def _sharded_worker(rank, world, port, q, bam, out_prefix):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), TIDDIT_HIP_DEVICE="0")
    import torch.distributed as dist
    from tiddit_amd import dist as tdist
    from tiddit_amd import tiddit_coverage
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the box's one GPU; the exchange runs over gloo
    try:
        header, cov, n = tdist.coverage_sharded(bam, 200, 10, chunk=1_500_000)
        if rank == 0:
            tiddit_coverage.print_coverage(cov, header, 200, "bed", out_prefix + ".bed")
        q.put((rank, n))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_coverage_sharded_multi_process(bams, tmp_path, world):
    """real ranks: every process ingests its byte range of ONE BAM on the device, the seams are checked with an all-gather and
    the bins meet in an all-reduce; rank 0's .bed equals the single-process CLI output and the record counts add up"""
    import socket
    import torch.multiprocessing as mp
    from tiddit_amd import __main__ as cli
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_sharded_worker, args=(r, world, port, q, bams[1], str(tmp_path / "sh"))) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(isinstance(v, int) for v in res.values()), res
    cli.run_cov(cli._cov_parser().parse_args(["--cov", "--bam", bams[1], "-o", str(tmp_path / "one"), "-z", "200", "-q", "10"]))
    assert open(str(tmp_path / "sh.bed")).read() == open(str(tmp_path / "one.bed")).read()
    want, _, _ = _host_records(bams[1])
    assert sum(res.values()) == len(want["tid"])


class _BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, v, n):                 # LSB first (header fields, extra bits)
        self.acc |= (v & ((1 << n) - 1)) << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):                 # Huffman codes are packed MSB first
        for i in range(n - 1, -1, -1):
            self.bits((c >> i) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xff)
        return bytes(self.out)


def _canon(lengths):
    """canonical code of every symbol from its length (RFC 1951 3.2.2)"""
    bl = [0] * 16
    for l in lengths:
        bl[l] += 1
    bl[0] = 0
    code, nxt = 0, [0] * 16
    for b in range(1, 16):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    out = []
    for l in lengths:
        out.append(nxt[l] if l else 0)
        if l:
            nxt[l] += 1
    return out


def _huff_lengths(weights):
    """code lengths of a Huffman code (complete prefix code) for {symbol: weight}, at least two symbols"""
    import heapq
    heap = [(w, i, (s,)) for i, (s, w) in enumerate(sorted(weights.items()))]
    heapq.heapify(heap)
    depth = {s: 0 for s in weights}
    tick = len(heap)
    while len(heap) > 1:
        w1, _, a = heapq.heappop(heap)
        w2, _, b = heapq.heappop(heap)
        for s in a + b:
            depth[s] += 1
        heapq.heappush(heap, (w1 + w2, tick, a + b))
        tick += 1
    return depth


_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DEXT = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]


def _dynamic_block(w, syms, ll_len, d_len, final):
    """one dynamic-Huffman DEFLATE block from (literal | (length, distance)) symbols with the GIVEN code lengths;
    the code-length sequence uses the 16/17/18 run codes (runs may cross from the literal/length into the distance lengths)"""
    hlit = max(257, max(i for i, l in enumerate(ll_len) if l) + 1)
    hdist = max(1, max([i for i, l in enumerate(d_len) if l] + [0]) + 1)
    seq = list(ll_len[:hlit]) + list(d_len[:hdist])
    rl, i = [], 0                                         # run-length code the sequence
    while i < len(seq):
        v, j = seq[i], i
        while j < len(seq) and seq[j] == v:
            j += 1
        run = j - i
        if v == 0 and run >= 3:
            r = min(run, 138)
            rl.append((18, r - 11, 7) if r >= 11 else (17, r - 3, 3))
            i += r
        elif v and run >= 4:
            rl.append((v, 0, 0))
            r = min(run - 1, 6)
            rl.append((16, r - 3, 2))
            i += 1 + r
        else:
            rl.append((v, 0, 0))
            i += 1
    freq = {}
    for s, _, _ in rl:
        freq[s] = freq.get(s, 0) + 1
    assert len(freq) >= 2
    cl = _huff_lengths(freq)
    if max(cl.values()) > 7:
        cl = _huff_lengths({k: 1 for k in freq})
    cl_len = [cl.get(s, 0) for s in range(19)]
    order19 = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    hclen = max(4, max(i for i, s in enumerate(order19) if cl_len[s]) + 1)
    w.bits(1 if final else 0, 1)
    w.bits(2, 2)
    w.bits(hlit - 257, 5)
    w.bits(hdist - 1, 5)
    w.bits(hclen - 4, 4)
    for s in order19[:hclen]:
        w.bits(cl_len[s], 3)
    cl_code = _canon(cl_len)
    for s, x, nb in rl:
        w.code(cl_code[s], cl_len[s])
        if nb:
            w.bits(x, nb)
    ll_code, d_code = _canon(ll_len), _canon(d_len)
    for s in syms + [256]:
        if isinstance(s, tuple):
            ln, dist = s
            lc = max(i for i in range(29) if _LBASE[i] <= ln)
            if lc == 28 and ln != 258:
                lc = 27
            w.code(ll_code[257 + lc], ll_len[257 + lc])
            w.bits(ln - _LBASE[lc], _LEXT[lc])
            dc = max(i for i in range(30) if _DBASE[i] <= dist)
            w.code(d_code[dc], d_len[dc])
            w.bits(dist - _DBASE[dc], _DEXT[dc])
        else:
            w.code(ll_code[s], ll_len[s])


def _skewed_lengths(symbols, n, maxlen=15):
    """a complete prefix code over `symbols` whose lengths run 1,2,3,...,maxlen,maxlen (then padded at maxlen)"""
    lens = [0] * n
    symbols = list(symbols)
    k = min(len(symbols), maxlen + 1)
    chain = list(range(1, k)) + [k - 1]
    for s, l in zip(symbols[:k], chain):
        lens[s] = l
    return lens


@pytest.mark.parametrize("kernel", ["lanes", "sequential"])
def test_device_inflate_hand_built_streams(ctx, kernel, monkeypatch):
    """DEFLATE streams zlib's encoder would never emit: 15-bit literal/length AND distance codes (canonical walk beyond the
    LUT on both tables), length 258 / distance 32768 and distance 1, a single distance code, code-length runs that cross from
    the literal/length into the distance lengths, empty stored blocks between Huffman blocks.  Validity is zlib's verdict."""
    if kernel == "sequential":
        monkeypatch.setenv("TIDDIT_INFLATE_SEQ", "1")
    rng = np.random.default_rng(21)
    streams = []
    # (1) sixteen literal/length symbols and sixteen distance symbols with weights 1,1,2,4,...: Huffman depths 15,15,14,...,1
    lits = [65 + i for i in range(13)]
    ll_syms = [256, 285, 257] + lits                         # EOB and the two length codes get the DEEPEST codes
    wts = [1, 1] + [2 ** i for i in range(1, 15)]
    lld = _huff_lengths(dict(zip(ll_syms, wts)))
    ll = [lld.get(s, 0) for s in range(286)]
    assert max(ll) == 15 and sorted(ll)[-2] == 15
    d_syms = [0, 29, 28, 26, 22, 18, 14, 10, 6, 1, 2, 3, 4, 24, 12, 16]
    dld = _huff_lengths(dict(zip(d_syms, wts)))
    dl = [dld.get(s, 0) for s in range(30)]
    assert max(dl) == 15
    data_syms = []
    out = bytearray()
    def emit(sym):
        data_syms.append(sym)
        if isinstance(sym, tuple):
            ln, dist = sym
            for _ in range(ln):
                out.append(out[-dist])
        else:
            out.append(sym)
    for _ in range(40000):
        emit(int(rng.choice(lits)))
    emit((258, 1))
    emit((3, 1))
    emit((258, 32768))
    emit((3, 24577))
    for dist in sorted(set([_DBASE[c] for c in d_syms] + [_DBASE[c] + (1 << _DEXT[c]) - 1 for c in d_syms])):   # both ends of every used code
        emit((3, dist))
        emit(lits[0])
    w = _BitWriter()
    _dynamic_block(w, data_syms, ll, dl, final=False)
    w.bits(0, 1); w.bits(0, 2)                               # an EMPTY stored block
    while w.n:
        w.bits(0, 1)
    w.bits(0, 16); w.bits(0xffff, 16)
    # (2) a block with ONE distance code (incomplete code) and literal lengths that end in a long zero run crossing into HDIST
    ll2 = [0] * 286
    for s in range(32, 64):
        ll2[s] = 6                                           # 32 symbols at depth 6 = half of the code space
    ll2[256] = 2
    ll2[257] = 2                                             # length 3; nothing above 257 -> zeros run on into the distance lengths
    dl2 = [0] * 30
    dl2[0] = 1
    syms2 = []
    base_len = len(out)
    for _ in range(5000):
        s = int(rng.integers(32, 64))
        syms2.append(s)
        out.append(s)
        if rng.random() < 0.2:
            syms2.append((3, 1))
            out.extend(out[-1:] * 3)
    _dynamic_block(w, syms2, ll2, dl2, final=True)
    raw = w.done()
    assert zlib.decompress(raw, -15) == bytes(out)           # the stream is valid DEFLATE and means what we think
    data = bytes(out)
    assert len(data) < 65536
    comp = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))) + bamio._BGZF_EOF
    assert _inflate_hbm(ctx, comp, len(data)) == data


def test_device_ingest_degenerate_files(ctx, tmp_path):
    """header-only BAM, a single record, only unplaced reads: the device reader and the CLI behave like the host reader"""
    from tiddit_amd import __main__ as cli
    refs = [("c1", 5000), ("c2", 3000)]
    paths = {}
    for kind in ("empty", "one", "unplaced"):
        p = str(tmp_path / (kind + ".bam"))
        w = bamio.BamWriter(p, refs)
        if kind == "one":
            w.write("r1", 99, 0, 100, 60, "50M", 0, 300, 250, "A" * 50)
        if kind == "unplaced":
            for i in range(500):
                w.write("u%d" % i, 77, -1, -1, 0, "", -1, -1, 0, "ACGT" * 10)
        w.close()
        paths[kind] = p
    for kind, p in paths.items():
        want, _, _ = (_host_records(p) if kind != "empty" else ({k: np.zeros(0) for k in FIELDS}, [], []))
        rd = bamio.DeviceBamReader(p, ctx=ctx)
        got = [b for b in rd.batches()]
        n = sum(len(b) for b in got)
        assert n == len(want["tid"]), kind
        if n:
            assert np.array_equal(np.concatenate([b.pos for b in got]), want["pos"]) and np.array_equal(np.concatenate([b.tid for b in got]), want["tid"])
        rd.close()
        cli.run_cov(cli._cov_parser().parse_args(["--cov", "--bam", p, "-o", str(tmp_path / ("o_" + kind)), "-z", "500"]))
        bed = open(str(tmp_path / ("o_" + kind)) + ".bed").read().splitlines()
        assert len(bed) == 1 + 10 + 6                                   # header + ceil(5000/500) + ceil(3000/500) rows
        if kind != "one":
            assert all(l.endswith("\t0.0") for l in bed[1:])
        else:
            assert bed[1].endswith("\t0.10000000149011612")               # 50 bases / 500 as float32


def _libdeflate():
    import ctypes
    try:
        L = ctypes.CDLL("libdeflate.so.0")
    except OSError:
        return None
    L.libdeflate_alloc_compressor.restype = ctypes.c_void_p
    L.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
    L.libdeflate_deflate_compress.restype = ctypes.c_size_t
    L.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.libdeflate_deflate_compress_bound.restype = ctypes.c_size_t
    L.libdeflate_deflate_compress_bound.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.libdeflate_free_compressor.argtypes = [ctypes.c_void_p]
    return L


@pytest.mark.parametrize("kernel", ["lanes", "sequential"])
def test_device_inflate_libdeflate_streams(ctx, kernel, monkeypatch):
    """htslib >= 1.10 writes BAM through libdeflate, whose DEFLATE streams are shaped differently from zlib's (block splitting,
    code-length choices, fixed blocks for short data): every level's output inflates to the same bytes on the device"""
    import ctypes
    L = _libdeflate()
    if L is None:
        pytest.skip("libdeflate.so.0 not present")
    if kernel == "sequential":
        monkeypatch.setenv("TIDDIT_INFLATE_SEQ", "1")
    cases = _cases()
    for level in (1, 3, 6, 9, 12):
        comp = L.libdeflate_alloc_compressor(level)
        for name, data in cases.items():
            out = b""
            for o in range(0, max(1, len(data)), 0xff00):
                d = data[o:o + 0xff00]
                cap = L.libdeflate_deflate_compress_bound(comp, len(d))
                buf = ctypes.create_string_buffer(cap)
                n = L.libdeflate_deflate_compress(comp, d, len(d), buf, cap)
                assert n > 0
                raw = buf.raw[:n]
                out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                        struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
            assert _inflate_hbm(ctx, out + bamio._BGZF_EOF, len(data)) == data, (name, level)
        L.libdeflate_free_compressor(comp)


@pytest.mark.parametrize("seed", range(12))
def test_device_ingest_random_legal_records(ctx, tmp_path, seed):
    """the record sanity checks behind the parallel record finder must accept every legal record: twelve random files, every field
    equal to the host decoder's, and not one batch handed to the host chase"""
    path = str(tmp_path / "rnd.bam")
    n = synth_bam.write_random_bam(path, 100 + seed)
    want, sa_w, _ = _host_records(path)
    got, sa_g, runs_ok, nb, hc = _device_records(path, ctx, int(np.random.default_rng(seed).choice([1 << 28, 70_000, 200_000])))
    assert len(want["tid"]) == n == len(got["tid"])
    for k in FIELDS:
        if k not in ("rec_off", "sa_off"):
            assert np.array_equal(want[k], got[k]), k
    assert sa_w == sa_g and runs_ok and hc == 0


def test_device_buffer_cache_is_bounded_flushed_and_can_be_switched_off(ctx, bams, tmp_path):
    """the readers' device buffers go to a per-device cache when a reader closes (GB-sized hipFree / hipMalloc bursts cost 10-40 ms each):
    a second reader takes them from there, tdt_device_cache_flush hands them back to the driver, and with TIDDIT_INGEST_CACHE_MB=0 (read
    once per process, hence the child) nothing is ever held"""
    import ctypes
    import subprocess
    import sys
    lib = ctx.lib
    _native.check(lib.tdt_device_cache_flush(ctx.handle, None))
    assert lib.tdt_device_cache_bytes(ctx.handle) == 0
    want, _, _, _, _ = _device_records(bams[1], ctx, 2_000_000)
    held = lib.tdt_device_cache_bytes(ctx.handle)
    assert held >= 1 << 20                                         # the closed reader's buffers
    got, _, _, _, _ = _device_records(bams[1], ctx, 2_000_000)     # ... serve the next reader (a buffer that had to grow leaves its old block behind)
    assert held <= lib.tdt_device_cache_bytes(ctx.handle) <= 2 * held
    held = lib.tdt_device_cache_bytes(ctx.handle)
    assert all(np.array_equal(want[k], got[k]) for k in FIELDS if k not in ("rec_off", "sa_off"))
    import torch
    free0, total = torch.cuda.mem_get_info()
    assert held <= total // 4                                      # the bound: a quarter of the device at most
    released = ctypes.c_uint64(0)
    _native.check(lib.tdt_device_cache_flush(ctx.handle, ctypes.byref(released)))
    assert released.value == held and lib.tdt_device_cache_bytes(ctx.handle) == 0
    assert torch.cuda.mem_get_info()[0] >= free0                   # (the driver has the memory back; how much of it shows as free at once is the runtime's business)
    # an allocation of the library that the driver refuses takes the cache with it and succeeds on the second try
    _device_records(bams[1], ctx, 2_000_000)
    assert lib.tdt_device_cache_bytes(ctx.handle) >= 1 << 20
    lib.tdt_debug_fail_next_malloc(1)
    from tiddit_amd import tiddit_coverage
    h = tiddit_coverage.CoverageHistogram([("c", 1_000_000)], 50, ctx=ctx)        # (its accumulators come from tdt_dev_malloc)
    assert lib.tdt_device_cache_bytes(ctx.handle) == 0
    h.close()
    lib.tdt_debug_fail_next_malloc(1)                              # ... and with nothing cached the refused allocation is still tried again:
    h = tiddit_coverage.CoverageHistogram([("c", 1_000_000)], 50, ctx=ctx)        # the hook cannot turn an allocation into a failure
    h.close()
    lib.tdt_debug_fail_next_malloc(0)
    code = ("import sys; sys.path.insert(0, %r); from tiddit_amd import _native, bamio\n"
            "c = _native.default_context()\n"
            "r = bamio.DeviceBamReader(%r, ctx=c, chunk=2000000); n = sum(len(b) for b in r.batches()); r.close()\n"
            "print(n, c.lib.tdt_device_cache_bytes(c.handle))") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), bams[1])
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TIDDIT_INGEST_CACHE_MB="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    n, cached = map(int, out.stdout.split()[-2:])
    assert n == len(want["pos"]) and cached == 0


@pytest.mark.parametrize("chunk", [300_000, 2_000_000])
def test_inflate_ahead_gives_the_same_batches(ctx, bams, chunk, monkeypatch):
    """Spans are inflated ahead of their turn (tdt_ingest_push_ahead: own stream, own output buffer, the carried record put in front
    afterwards): the batches equal those of a reader that never starts a span early, the current batch — raw bytes included — stays
    readable while the next span inflates, and ahead() starts a span once"""
    monkeypatch.setenv("TIDDIT_INGEST_AHEAD", "0")
    want, _, _, nb, _ = _device_records(bams[1], ctx, chunk)
    monkeypatch.delenv("TIDDIT_INGEST_AHEAD")
    r = bamio.DeviceBamReader(bams[1], ctx=ctx, chunk=chunk)
    cols = {k: [] for k in FIELDS}
    for b in r.batches():
        first = b.pos[:4].copy()                                       # (a host copy made BEFORE the call)
        r.ahead()
        assert r.ahead() is False                                      # one span beyond the current one
        i = len(b) - 1                                                 # the raw bytes are still the batch's own
        assert b.raw[int(b.rec_off[i]) + 4:int(b.rec_off[i]) + 8].view(np.int32)[0] == b.tid[i]
        for k in FIELDS:
            if k not in ("rec_off", "sa_off"):
                cols[k].append(getattr(b, k))                          # ... the field arrays are copied AFTER it
        assert np.array_equal(first, cols["pos"][-1][:4])
    assert r.spans_ahead >= nb // 2 or nb <= 2                         # (the reader thread does not always have the next span yet)
    r.close()
    for k in cols:
        if k not in ("rec_off", "sa_off"):                             # (batch-relative)
            assert np.array_equal(np.concatenate(cols[k]), want[k]), k


@pytest.mark.parametrize("gap", ["0", "256", "65536"])
def test_carried_record_longer_than_the_gap_is_relocated(ctx, bams, gap, monkeypatch):
    """the span's output starts `gap` bytes into its buffer and the partial record of the batch before is copied in front of it; a record
    longer than the gap (here: a gap of 0 / 256 bytes / 64 KB on ordinary records) takes the relocation path — same records either way"""
    want, sa_w, _, _, _ = _device_records(bams[0], ctx, 1 << 28)
    monkeypatch.setenv("TIDDIT_INGEST_GAP", gap)
    got, sa_g, runs_ok, nb, hc = _device_records(bams[0], ctx, 150_000)
    assert nb > 3 and runs_ok and hc == 0 and sa_g == sa_w
    for k in FIELDS:
        if k not in ("rec_off", "sa_off"):
            assert np.array_equal(got[k], want[k]), k


def test_retained_batches_survive_spans_inflating_ahead(ctx, bams):
    """a retained batch owns its buffers: later spans — begun ahead or not — inflate elsewhere, and every retained batch still reads back
    as it was when the reader has finished the file"""
    want, _, _, _, _ = _device_records(bams[1], ctx, 1 << 28)
    r = bamio.DeviceBamReader(bams[1], ctx=ctx, chunk=400_000)
    r.retain = True
    kept = []
    for b in r.batches():
        r.ahead()
        kept.append(b)
    assert len(kept) > 4 and r.spans_ahead > 0
    pos = np.concatenate([b.pos for b in kept])
    flag = np.concatenate([b.flag for b in kept])
    for b in kept:
        i = len(b) // 3
        assert b.raw[int(b.rec_off[i]) + 4:int(b.rec_off[i]) + 8].view(np.int32)[0] == b.tid[i]
        b.release()
    r.close()
    assert np.array_equal(pos, want["pos"]) and np.array_equal(flag, want["flag"])


def test_inflate_ahead_misuse_is_refused(ctx, bams):
    """a push for another span than the one begun first, and a third span begun beyond the current batch, are errors"""
    import ctypes
    lib = ctx.lib
    r = bamio.DeviceBamReader(bams[1], ctx=ctx, chunk=300_000)
    it = r.batches()
    b = next(it)
    assert len(b) > 0
    for _ in range(200):                                               # until the reader thread has the next span
        if r.spans_ahead or r.ahead():
            break
        import time
        time.sleep(0.01)
    assert r.spans_ahead == 1
    buf = np.zeros(64, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    span = r._gen["pending"]
    assert lib.tdt_ingest_push_ahead(r._h, _native.ptr(span[0]), span[1]) == 0          # a second span beyond the current batch: fine
    assert lib.tdt_ingest_push_ahead(r._h, _native.ptr(span[0]), span[1]) != 0 and b"two spans are already" in lib.tdt_last_error()
    assert lib.tdt_ingest_push(r._h, _native.ptr(buf), 64, 0, ctypes.byref(n)) != 0
    assert b"another span was started" in lib.tdt_last_error()
    it.close()
    r.close()
