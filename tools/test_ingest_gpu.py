"""Device ingest (inflate + record finding + field decode) vs the host decoder (run on the GPU box)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, bamio, synth_bam

ctx = _native.default_context(); lib = ctx.lib
NAMES = [("tid", np.int32), ("pos", np.int32), ("end", np.int32), ("mapq", np.uint8), ("flag", np.uint16), ("mate_tid", np.int32),
         ("mate_pos", np.int32), ("tlen", np.int32), ("l_seq", np.int32), ("cigar_first", np.uint32), ("cigar_last", np.uint32),
         ("rec_off", np.uint64), ("sa_off", np.int64)]


def host_all(path):
    r = bamio.BamReader(path)
    out = {k: [] for k, _ in NAMES}
    sa = []
    for b in r.batches():
        for k, _ in NAMES:
            out[k].append(getattr(b, k))
        sa.extend(b.record(i).get_tag_sa() for i in np.flatnonzero(b.sa_off >= 0))
    r.close()
    return r, {k: np.concatenate(v) for k, v in out.items()}, sa


def header_len(path):
    """inflated bytes in front of the first record"""
    import struct
    blocks = bamio.bgzf_blocks(open(path, "rb"))
    buf = b""
    def need(n):
        nonlocal buf
        while len(buf) < n:
            buf += next(blocks)
    need(12)
    l_text = struct.unpack_from("<i", buf, 4)[0]
    need(12 + l_text)
    o = 8 + l_text
    nref = struct.unpack_from("<i", buf, o)[0]; o += 4
    for _ in range(nref):
        need(o + 4)
        ln = struct.unpack_from("<i", buf, o)[0]
        need(o + 8 + ln)
        o += 4 + ln + 4
    return o, nref


def device_all(path, chunk):
    skip, nref = header_len(path)
    comp = np.fromfile(path, dtype=np.uint8)
    h = ctypes.c_void_p()
    _native.check(lib.tdt_ingest_create(ctx.handle, nref, ctypes.byref(h)))
    out = {k: [] for k, _ in NAMES}
    sa = []
    o = 0
    nedges = 0
    while o < len(comp):
        nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        part = comp[o:o + chunk]
        _native.check(lib.tdt_bgzf_scan(_native.ptr(part), len(part), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
        assert consumed.value > 0
        n = ctypes.c_size_t(0)
        _native.check(lib.tdt_ingest_push(h, _native.ptr(part), consumed.value, skip if o == 0 else 0, ctypes.byref(n)))
        o += consumed.value
        ptrs = (ctypes.c_void_p * 14)()
        rawlen = ctypes.c_size_t(0)
        _native.check(lib.tdt_ingest_arrays(h, ptrs, ctypes.byref(rawlen)))
        n = n.value
        if not n:
            continue
        batch = {}
        for i, (k, dt) in enumerate(NAMES):
            a = np.empty(n, dtype=dt)
            _native.check(lib.tdt_copy_to_host(ctx.handle, _native.ptr(a), ptrs[i], a.nbytes))
            batch[k] = a
            out[k].append(a)
        raw = np.empty(rawlen.value, dtype=np.uint8)
        _native.check(lib.tdt_copy_to_host(ctx.handle, _native.ptr(raw), ptrs[13], raw.nbytes))
        for i in np.flatnonzero(batch["sa_off"] >= 0):
            s = int(batch["sa_off"][i]); e = s
            while raw[e]:
                e += 1
            sa.append(bytes(raw[s:e]).decode())
        # rec_off really points at the record
        k = n // 2
        assert raw[int(batch["rec_off"][k]) + 4:int(batch["rec_off"][k]) + 8].view(np.int32)[0] == batch["tid"][k]
        ed = np.empty(1024, dtype=np.uint32); ne = ctypes.c_size_t(0)
        _native.check(lib.tdt_ingest_edges(h, _native.ptr(ed), 1024, ctypes.byref(ne)))
        want_ed = np.concatenate([[0], np.flatnonzero(np.diff(batch["tid"])) + 1])
        assert np.array_equal(ed[:ne.value], want_ed), (ed[:ne.value], want_ed)
        nedges += ne.value
    c = ctypes.c_size_t(0)
    hc = ctypes.c_size_t(0)
    _native.check(lib.tdt_ingest_carry(h, ctypes.byref(c), ctypes.byref(hc)))
    assert c.value == 0, c.value
    global HOST_CHASES
    HOST_CHASES += hc.value
    lib.tdt_ingest_destroy(h)
    return {k: np.concatenate(v) for k, v in out.items()}, sa


def compare(path, chunk):
    _, want, sa_w = host_all(path)
    got, sa_g = device_all(path, chunk)
    ok = True
    for k, _ in NAMES:
        if k in ("rec_off", "sa_off"):
            continue      # batch-relative offsets: batches differ between the two readers
        if not np.array_equal(want[k], got[k]):
            ok = False
            print("  MISMATCH", k, len(want[k]), len(got[k]))
    if sa_w != sa_g:
        ok = False
        print("  SA strings differ", len(sa_w), len(sa_g))
    print(os.path.basename(path), "chunk", chunk, "records", len(want["tid"]), "SA", len(sa_w), "OK" if ok else "FAILED")
    return ok


HOST_CHASES = 0
d = "/tmp/ingest_t"; os.makedirs(d, exist_ok=True)
sv = d + "/sv.bam"
synth_bam.write_synthetic_bam(sv, [("chr1", 300000), ("chr2", 200000), ("chrM", 3000), ("tiny", 500)], depth=8, seed=5)
bulk = d + "/bulk.bam"
synth_bam.write_bulk_bam(bulk, [("chr1", 3_000_000), ("chr2", 2_000_000)], depth=30, threads=16)
allok = True
for path in (sv, bulk):
    for chunk in (1 << 30, 3_000_000, 400_000):
        allok &= compare(path, chunk)
print("ALL OK" if allok else "FAILURES", "host chases:", HOST_CHASES)
