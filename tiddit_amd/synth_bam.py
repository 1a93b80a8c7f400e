"""Synthetic coordinate-sorted paired-end BAMs with planted structural variants (test / config
tooling; the reference ships no sample data).  Everything is derived from a seed."""
import numpy as np

from .bamio import BamWriter

_BASES = "ACGT"


def write_synthetic_bam(path, contigs, depth=10, read_len=100, insert=350, insert_sd=30, seed=1, n_events=12,
                        sample="SYN", with_rg=True):
    """contigs: list of (name, length).  -> dict with the planted events.
    Planted: deletions (pairs with a large insert + split reads with SA tags), inter-contig translocations
    (mates on different contigs), inversions (same-orientation pairs), soft-clipped reads, plus
    duplicates, low-mapq, secondary/supplementary records, unpaired and mate-unmapped reads."""
    rng = np.random.default_rng(seed)
    recs = []   # (tid, pos, dict)
    big = [i for i, (n, l) in enumerate(contigs) if l > 16 * insert]

    def seq(n):
        return "".join(_BASES[i] for i in rng.integers(0, 4, n))

    def pair(name, tidA, posA, tidB, posB, revA=False, revB=True, mapqA=60, mapqB=60, extra_flagA=0, extra_flagB=0,
             cigA=None, cigB=None, tagsA=(), tagsB=()):
        cigA = cigA or "%dM" % read_len
        cigB = cigB or "%dM" % read_len
        tl = (posB + read_len - posA) if tidA == tidB else 0
        fa = 0x1 | 0x40 | (0x10 if revA else 0) | (0x20 if revB else 0) | extra_flagA
        fb = 0x1 | 0x80 | (0x10 if revB else 0) | (0x20 if revA else 0) | extra_flagB
        if tidA == tidB and abs(tl) < 3 * insert and not revA and revB:
            fa |= 0x2
            fb |= 0x2
        recs.append((tidA, posA, dict(qname=name, flag=fa, tid=tidA, pos=posA, mapq=mapqA, cigar=cigA, mate_tid=tidB, mate_pos=posB,
                                      tlen=tl, seq=seq(read_len), tags=tagsA)))
        recs.append((tidB, posB, dict(qname=name, flag=fb, tid=tidB, pos=posB, mapq=mapqB, cigar=cigB, mate_tid=tidA, mate_pos=posA,
                                      tlen=-tl, seq=seq(read_len), tags=tagsB)))

    q = 0
    for tid, (name, L) in enumerate(contigs):
        if L < 3 * insert:
            n_pairs = 2
        else:
            n_pairs = int(L * depth / (2 * read_len))
        for _ in range(n_pairs):
            q += 1
            ins = max(read_len + 1, int(rng.normal(insert, insert_sd)))
            if L <= ins + 2:
                posA, posB = 0, max(0, L - read_len)
            else:
                posA = int(rng.integers(0, L - ins))
                posB = posA + ins - read_len
            u = rng.random()
            kw = {}
            if u < 0.02:
                kw["extra_flagA"] = 0x400
            elif u < 0.05:
                kw["mapqA"] = int(rng.integers(0, 5))
            elif u < 0.06:
                kw["extra_flagB"] = 0x100
            elif u < 0.07:
                kw["extra_flagB"] = 0x800
            elif u < 0.10 and L > 4 * read_len:
                c = int(rng.integers(26, 40))
                kw["cigA"] = "%dS%dM" % (c, read_len - c) if rng.random() < 0.5 else "%dM%dS" % (read_len - c, c)
            elif u < 0.11:
                kw["cigA"] = "40M5D60M"
            pair("p%d" % q, tid, posA, tid, posB, **kw)
    events = []
    for ev in range(n_events):
        kind = ["DEL", "BND", "INV", "DEL"][ev % 4]
        tA = big[int(rng.integers(0, len(big)))]
        LA = contigs[tA][1]
        a = int(rng.integers(2 * insert, LA - 12 * insert))
        support = int(rng.integers(4, 10))
        if kind == "DEL":
            size = int(rng.integers(3 * insert, 8 * insert))
            b = a + size
            for k in range(support):
                q += 1
                pair("del%d_%d" % (ev, k), tA, a - int(rng.integers(read_len, insert)), tA, b + int(rng.integers(0, insert - read_len)))
            for k in range(int(rng.integers(3, 7))):      # split reads across the junction
                q += 1
                left = int(rng.integers(35, 65))
                sa = "%s,%d,+,%dS%dM,60,0;" % (contigs[tA][0], b + 1, left, read_len - left)
                pair("dsp%d_%d" % (ev, k), tA, a - left, tA, a - left + insert - read_len, cigA="%dM%dS" % (left, read_len - left),
                     tagsA=[("SA", "Z", sa), ("NM", "i", 0)])
            events.append({"type": "DEL", "chrom": contigs[tA][0], "start": a, "end": b})
        elif kind == "BND":
            tB = big[int(rng.integers(0, len(big)))]
            while tB == tA and len(big) > 1:
                tB = big[int(rng.integers(0, len(big)))]
            LB = contigs[tB][1]
            b = int(rng.integers(2 * insert, LB - 2 * insert))
            for k in range(support):
                q += 1
                pair("bnd%d_%d" % (ev, k), tA, a - int(rng.integers(read_len, insert)), tB, b + int(rng.integers(0, insert - read_len)))
            for k in range(int(rng.integers(0, 5))):
                q += 1
                left = int(rng.integers(35, 65))
                sa = "%s,%d,%s,%dS%dM,%d,1;%s,%d,+,50M50S,0,0;" % (contigs[tB][0], b + 1, "-" if k % 2 else "+", left, read_len - left,
                                                                     60 if k != 1 else 2, contigs[tA][0], 5)
                pair("bsp%d_%d" % (ev, k), tA, a - left, tA, a - left + insert - read_len, cigA="%dM%dS" % (left, read_len - left),
                     tagsA=[("SA", "Z", sa)])
            events.append({"type": "BND", "chromA": contigs[tA][0], "posA": a, "chromB": contigs[tB][0], "posB": b})
        else:
            size = int(rng.integers(4 * insert, 9 * insert))
            b = a + size
            for k in range(support):
                q += 1
                pair("inv%d_%d" % (ev, k), tA, a - int(rng.integers(read_len, insert)), tA, b - int(rng.integers(read_len, insert)),
                     revA=False, revB=False)
            events.append({"type": "INV", "chrom": contigs[tA][0], "start": a, "end": b})
    # odd records: mate unmapped, unpaired, unmapped-but-placed
    for k in range(10):
        tA = big[int(rng.integers(0, len(big)))]
        p = int(rng.integers(0, contigs[tA][1] - read_len))
        recs.append((tA, p, dict(qname="mu%d" % k, flag=0x1 | 0x8 | 0x40, tid=tA, pos=p, mapq=60, cigar="%dM" % read_len, mate_tid=tA,
                                 mate_pos=p, tlen=0, seq=seq(read_len), tags=())))
        recs.append((tA, p, dict(qname="mu%d" % k, flag=0x1 | 0x4 | 0x80, tid=tA, pos=p, mapq=0, cigar="", mate_tid=tA, mate_pos=p,
                                 tlen=0, seq=seq(read_len), tags=())))
        recs.append((tA, p + 7, dict(qname="se%d" % k, flag=0, tid=tA, pos=p + 7, mapq=60, cigar="%dM" % read_len, mate_tid=-1,
                                     mate_pos=-1, tlen=0, seq=seq(read_len), tags=())))
    order = sorted(range(len(recs)), key=lambda i: (recs[i][0], recs[i][1]))   # stable
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    if with_rg:
        text += "@RG\tID:rg1\tSM:%s\n" % sample
    w = BamWriter(path, contigs, text=text)
    for i in order:
        w.write(**recs[i][2])
    w.close()
    return {"events": events, "n_records": len(recs)}


def write_bulk_bam(path, contigs, depth=30, read_len=100, insert=350, seed=1, level=1, threads=8, chunk=1 << 20, realistic=False):
    """Large plain paired-end BAM written with numpy (fixed-size records: 12-byte name, one M cigar op, no aux), for
    end-to-end timing of the BGZF/BAM ingest.  ~10 M records/min.  -> number of records.
    realistic: reads are cut from one random reference per contig (overlapping reads share sequence, as in a real
    coordinate-sorted BAM, so DEFLATE finds long matches) and base qualities come in runs; the default draws every base and
    quality independently (hard to compress, short matches)."""
    import struct
    from concurrent.futures import ThreadPoolExecutor
    from .bamio import _BGZF_EOF, _bgzf_block
    rng = np.random.default_rng(seed)
    rec = np.dtype([("block_size", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                    ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("mate_tid", "<i4"), ("mate_pos", "<i4"), ("tlen", "<i4"),
                    ("name", "S12"), ("cigar", "<u4"), ("seq", "u1", ((read_len + 1) // 2,)), ("qual", "u1", (read_len,))])
    text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in contigs)).encode()
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs))
    for name, ln in contigs:
        nb = name.encode() + b"\x00"
        head += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
    total = 0
    qual_lut = np.sort(np.minimum(40, 2 + (38 * np.sqrt(np.arange(256) / 255.0)).astype(np.int64)).astype(np.uint8))   # skewed to high quality
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as pool:
        pend = bytearray(head)

        def flush(final=False):
            nonlocal pend
            nblk = len(pend) // 0xff00 if not final else -(-len(pend) // 0xff00)
            view = bytes(pend[:nblk * 0xff00])
            for blk in pool.map(lambda o: _bgzf_block(view[o:o + 0xff00], level), range(0, len(view), 0xff00)):
                f.write(blk)
            del pend[:nblk * 0xff00]

        for tid, (name, L) in enumerate(contigs):
            ref_nib = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, L + read_len + 2, dtype=np.uint8)] if realistic else None
            n_pairs = int(L * depth / (2 * read_len))
            posA = rng.integers(0, max(1, L - insert - 100), n_pairs).astype(np.int64)
            ins = np.maximum(read_len + 1, rng.normal(insert, 30, n_pairs).astype(np.int64))
            posB = np.minimum(posA + ins - read_len, L - read_len)
            pos = np.concatenate([posA, posB])
            mate = np.concatenate([posB, posA])
            tl = np.concatenate([posB + read_len - posA, -(posB + read_len - posA)])
            flag = np.concatenate([np.full(n_pairs, 0x1 | 0x2 | 0x40 | 0x20), np.full(n_pairs, 0x1 | 0x2 | 0x80 | 0x10)])
            order = np.argsort(pos, kind="stable")
            pos, mate, tl, flag = pos[order], mate[order], tl[order], flag[order]
            for lo in range(0, len(pos), chunk):
                hi = min(len(pos), lo + chunk)
                m = hi - lo
                a = np.zeros(m, dtype=rec)
                a["block_size"] = rec.itemsize - 4
                a["tid"], a["pos"], a["l_name"] = tid, pos[lo:hi], 12
                u = rng.random(m)
                a["mapq"] = np.where(u < 0.05, rng.integers(0, 20, m), 60)
                a["n_cigar"], a["l_seq"] = 1, read_len
                a["flag"] = flag[lo:hi] | np.where(rng.random(m) < 0.02, 0x400, 0)
                a["mate_tid"], a["mate_pos"], a["tlen"] = tid, mate[lo:hi], tl[lo:hi]
                a["bin"] = 4681 + (pos[lo:hi] >> 14)
                ids = (total + order[lo:hi] % max(1, n_pairs)).astype(np.int64)
                digits = (ids[:, None] // (10 ** np.arange(10, -1, -1, dtype=np.int64))[None, :] % 10 + 48).astype(np.uint8)
                a["name"] = np.concatenate([digits, np.zeros((m, 1), np.uint8)], axis=1).view("S12")[:, 0]
                a["cigar"] = (read_len << 4) | 0
                if realistic:
                    idx = pos[lo:hi, None] + np.arange(read_len + (read_len & 1))[None, :]
                    nib = ref_nib[idx]
                    mism = rng.random(nib.shape) < 0.003                      # sequencing errors / variants
                    nib = np.where(mism, np.roll(nib, 1, axis=1), nib)
                    q = np.repeat(qual_lut[rng.integers(96, 256, (m, (read_len + 9) // 10), dtype=np.uint8)], 10, axis=1)[:, :read_len]
                    a["qual"] = np.where(rng.random((m, read_len)) < 0.1, qual_lut[rng.integers(0, 256, (m, read_len), dtype=np.uint8)], q)
                else:
                    nib = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, (m, read_len + (read_len & 1)), dtype=np.uint8)]
                    a["qual"] = qual_lut[rng.integers(0, 256, (m, read_len), dtype=np.uint8)]
                a["seq"] = (nib[:, 0::2] << 4) | nib[:, 1::2]
                pend += a.tobytes()
                flush()
            total += len(pos)
        flush(final=True)
        f.write(_BGZF_EOF)
    return total


def write_random_bam(path, seed):
    """records of every legal shape: names of 1..250 printable characters, CIGARs with all nine operations, empty sequences,
    unmapped reads, aux fields of every type (arrays included) around an optional SA:Z"""
    rng = np.random.default_rng(seed)
    refs = [("r%d" % i, int(rng.integers(1000, 3_000_000))) for i in range(int(rng.integers(1, 40)))]
    w = BamWriter(path, refs, level=int(rng.integers(1, 7)), align_records=bool(rng.integers(0, 2)))
    printable = [chr(c) for c in range(33, 127) if chr(c) != "@"]
    n = int(rng.integers(200, 3000))
    pos, tid = 0, 0
    for i in range(n):
        if rng.random() < 0.02 and tid + 1 < len(refs):
            tid, pos = tid + 1, 0
        pos = min(refs[tid][1] - 1, pos + int(rng.integers(0, 50)))
        name = "".join(rng.choice(printable, int(rng.choice([1, 2, 5, 20, 40, 100, 250]))))
        kind = rng.random()
        if kind < 0.05:                                                 # unmapped, no CIGAR
            cig, lseq, flag, t, p = [], int(rng.integers(0, 200)), 4 | 1, -1 if rng.random() < 0.5 else tid, -1 if rng.random() < 0.5 else pos
        else:
            ops = []
            for _ in range(int(rng.integers(1, 12))):
                ops.append((int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8])), int(rng.integers(1, 300))))
            lseq = sum(l for op, l in ops if op in (0, 1, 4, 7, 8))
            if rng.random() < 0.05:
                lseq = 0                                                # secondary alignments may omit the sequence
            cig, flag, t, p = ops, int(rng.choice([0, 16, 99, 147, 256, 2048 + 16, 1024 + 83])), tid, pos
        tags = []
        for _ in range(int(rng.integers(0, 6))):
            ty = str(rng.choice(["A", "c", "C", "s", "S", "i", "I", "f", "Z", "H", "BC", "Bs", "Bi", "Bf"]))
            tg = "X" + str(rng.choice(list("ABCDEFGH")))
            val = {"A": "q", "c": -5, "C": 200, "s": -3000, "S": 60000, "i": -70000, "I": 4000000000, "f": 1.5, "Z": "some text", "H": "1AE301",
                   "BC": [1, 2, 3], "Bs": [-1, 5], "Bi": list(range(int(rng.integers(0, 40)))), "Bf": [0.5]}[ty]
            tags.append((tg, ty, val))
        if rng.random() < 0.1:
            tags.insert(int(rng.integers(0, len(tags) + 1)), ("SA", "Z", "r0,%d,+,30M70S,%d,1;" % (pos + 5, int(rng.integers(0, 60)))))
        w.write(name, flag, t, p, int(rng.integers(0, 256)), cig, t if rng.random() < 0.8 else -1, int(rng.integers(-1, 1000)),
                int(rng.integers(-5000, 5000)), "".join(rng.choice(list("ACGTN"), lseq)), tuple(tags))
    w.close()
    return n
