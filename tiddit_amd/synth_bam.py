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


# ------------------------------------------------------------------------------------------------------------------
# WGS-shaped paired-end BAM with planted structural variants (BASELINE configs[3]: `tiddit --sv --skip_assembly`)
HUMAN_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]   # GRCh38 chr1..22, X, Y


def wgs_contigs(total_mb, decoys=True):
    """24 chromosomes with GRCh38's relative lengths scaled to `total_mb` in all (lengths are not multiples of any bin size),
    plus chrM and two unplaced scaffolds shorter than tiddit's default --min_contig (10000)."""
    scale = total_mb / float(sum(HUMAN_MB))
    names = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
    contigs = [(n, int(mb * scale * 1e6) // 1000 * 1000 + 137 + 11 * i) for i, (n, mb) in enumerate(zip(names, HUMAN_MB))]
    if decoys:
        contigs += [("chrM", 16569), ("chrUn_KI270302v1", 2274), ("chrUn_KI270304v1", 2165)]
    return contigs


def grch38_shaped_contigs(total_mb=24, seed=38):
    """A contig table with the SHAPE of the GRCh38 analysis set's header (GRCh38_full_analysis_set_plus_decoy_hla: 3 366 @SQ lines),
    scaled to about `total_mb` Mb: 24 chromosomes with GRCh38's relative lengths, chrM, 42 `*_random` and 127 `chrUn_*` scaffolds
    (970 bp up), 261 `*_alt` contigs, chrEBV, 2 385 `chrUn_JTFH...v1_decoy` contigs of 1-8 kb and 525 HLA alleles whose names carry
    `*` and `:` (`HLA-DRB1*15:03:01:02`; a few of them longer than tiddit's default --min_contig).  The names follow the real
    header's patterns (the accessions are synthetic); everything derives from `seed`.  What it is for: the header alone is larger
    than three BGZF blocks, thousands of contigs lie below --min_contig, hundreds of small contigs carry reads, the string order of
    the names (chrA < chrB decisions, tiddit_signal.pyx:213) differs from the header order, and the file ends in a tid = -1 tail."""
    rng = np.random.default_rng(seed)
    small = 0.5 * total_mb * 1e6
    prim_scale = (total_mb * 1e6 - small) / float(sum(HUMAN_MB))
    chrom = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
    out = [(n, int(mb * prim_scale) // 1000 * 1000 + 137 + 11 * i) for i, (n, mb) in enumerate(zip(chrom, HUMAN_MB))]
    out.append(("chrM", 16569))
    host = [1, 1, 1, 2, 2, 3, 4, 5, 9, 9, 9, 9, 11, 14, 14, 14, 14, 14, 14, 14, 14, 15, 16, 17, 17, 17, 22, 22, 22, 22, 22, 22, 22, 22, 22]
    for i in range(42):
        c = "Y" if i >= 41 else str(host[i % len(host)])
        out.append(("chr%s_KI270%03dv1_random" % (c, 706 + i), int(rng.integers(1200, 42000))))
    for i in range(127):
        ln = 970 if i == 0 else int(rng.integers(980, 1500)) if i < 12 else int(rng.integers(1500, 26000))
        out.append((("chrUn_KI270%03dv1" % (302 + i)) if i < 100 else ("chrUn_GL000%03dv1" % (195 + i - 100)), ln))
    for i in range(261):
        c = chrom[int(rng.integers(0, 24))][3:]
        out.append((("chr%s_KI270%03dv1_alt" % (c, 762 + i)) if i % 3 else ("chr%s_GL000%03dv2_alt" % (c, 250 + i // 3)), int(rng.integers(3000, 14000))))
    out.append(("chrEBV", 17182))
    for i in range(2385):
        out.append(("chrUn_JTFH0100%04dv1_decoy" % (i + 1), int(rng.integers(1000, 3600 if i % 7 else 8000))))
    genes = ["A", "B", "C", "DQA1", "DQB1", "DRB1"]
    for i in range(525):
        g = genes[i * len(genes) // 525]
        name = "HLA-%s*%02d:%02d:%02d:%02d" % (g, 1 + i % 57, 1 + (i // 3) % 40, 1 + i % 4, 1 + i % 3)
        if i % 11 == 0:
            name = name.rsplit(":", 1)[0] + ("N" if i % 22 == 0 else "")            # three-field names and null alleles (`HLA-A*01:11N`-like)
        ln = int(rng.integers(10500, 13900)) if g == "DRB1" and i % 4 == 0 else int(rng.integers(970, 5600))
        out.append((name, ln))
    seen = set()
    uniq = []
    for n, ln in out:                                           # (the generated allele names collide now and then: keep the header's names unique)
        k = 2
        base = n
        while n in seen:
            n = "%s:%02d" % (base, k)
            k += 1
        seen.add(n)
        uniq.append((n, ln))
    return uniq


def contigs_for(params):
    """the contig table of a synthetic WGS file from its parameter dictionary (tests/golden fixtures, bench.py): `contig_table` =
    "grch38" selects :func:`grch38_shaped_contigs`, otherwise :func:`wgs_contigs`"""
    if params.get("contig_table") == "grch38":
        return grch38_shaped_contigs(params["total_mb"])
    return wgs_contigs(params["total_mb"])


def write_fasta(path, contigs, seed=100, width=60):
    """reference FASTA for `contigs` (synth.gen_sequence per contig) -> {name: uint8 sequence}"""
    from .synth import gen_sequence
    seqs = {}
    with open(path, "wb") as f:
        for i, (name, L) in enumerate(contigs):
            s = gen_sequence(L, seed=seed + i)
            seqs[name] = s
            f.write((">%s\n" % name).encode())
            nfull = L // width
            body = np.empty((nfull, width + 1), dtype=np.uint8)
            body[:, :width] = s[:nfull * width].reshape(nfull, width)
            body[:, width] = 10
            f.write(body.tobytes())
            if L > nfull * width:
                f.write(s[nfull * width:].tobytes() + b"\n")
    return seqs


_NIB = np.zeros(256, dtype=np.uint8) + 15
for _c, _v in zip(b"=ACMGRSVTWYHKDBN", range(16)):
    _NIB[_c] = _v
    _NIB[_c | 0x20] = _v


def write_wgs_sv_bam(path, contigs, depth=30, read_len=150, insert=400, insert_sd=40, seed=7, sv_per_mb=3.0, noise_pair_frac=1e-3,
                     level=1, threads=8, sample="WGS", ref_seqs=None, chunk=1 << 18):
    """Coordinate-sorted paired-end BAM shaped like a 30x short-read WGS run, everything derived from `seed`:
      * FR pairs at `depth` over every contig (98 % of the records; fixed-size records built with numpy), 2 % duplicates,
        12 % of the reads below mapq 60, a random half of the pairs with read 2 leftmost;
      * per-pair variations written record by record: soft-clipped reads (clip candidates of tiddit_signal.worker :190-197),
        reads with a deletion in the CIGAR (reference_end > pos + l_seq), supplementary (SA-tagged, 0x800) and secondary (0x100)
        extra alignments, pairs with an unmapped mate;
      * planted events at `sv_per_mb` per Mb — deletions, tandem duplications, inversions (both junctions), inter-chromosomal
        translocations — each with 4..14 discordant pairs and 0..8 split reads whose primary carries an SA tag and whose
        supplementary alignment is present as its own record; a fifth of the SA tags list a second alignment, some point at
        low-mapq alignments, some fragments have SA tags on both mates (one qname, two split rows);
      * noise: `noise_pair_frac` of the pairs have the mate far away on the same contig or on another contig;
      * a tail of unplaced unmapped pairs.
    ref_seqs: {name: uint8 ASCII sequence} — reads are then cut from it (as an aligner would report them); otherwise random bases.
    -> dict(events=[...], n_records=...)"""
    import os
    import struct
    from concurrent.futures import ThreadPoolExecutor
    from .bamio import _BGZF_EOF, _bgzf_block, _reg2bin
    rl = read_len
    ncon = len(contigs)
    lens = np.array([l for _, l in contigs], dtype=np.int64)
    span = insert + 6 * insert_sd
    big = [t for t in range(ncon) if lens[t] > 60 * span]
    n_pairs = [int(l * depth / (2 * rl)) if l > 4 * span else 0 for l in lens]
    pair_base = np.concatenate([[0], np.cumsum(n_pairs)])
    specials = [[] for _ in range(ncon + 1)]          # per tid (last: unplaced): (pos, record bytes)
    erng = np.random.default_rng([seed, 1 << 20])
    pool = np.array([1, 2, 4, 8], np.uint8)[erng.integers(0, 4, 1 << 16, dtype=np.uint8)]
    pool = ((pool[0::2] << 4) | pool[1::2]).tobytes()
    qpool = np.minimum(40, 2 + (38 * np.sqrt(erng.integers(60, 256, 1 << 15) / 255.0)).astype(np.int64)).astype(np.uint8).tobytes()
    half = (rl + 1) // 2

    def packed_seq(tid, pos, n, rng):
        """4-bit packed bases of n query bases starting at reference pos (no reference: random)"""
        if ref_seqs is not None and tid >= 0:
            s = ref_seqs[contigs[tid][0]]
            seg = s[max(0, pos):max(0, pos) + n]
            nib = _NIB[seg]
            if len(nib) < n + (n & 1):
                nib = np.concatenate([nib, np.full(n + (n & 1) - len(nib), 1, np.uint8)])
            nib = nib[:n + (n & 1)].copy()
            if n & 1:
                nib[n] = 0
            return ((nib[0::2] << 4) | nib[1::2]).tobytes()
        o = int(rng.integers(0, len(pool) - half - 1))
        return pool[o:o + (n + 1) // 2]

    def enc(rng, qname, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, aux=b"", lseq=rl):
        """cigar: list of (op, len) (hard clips do not count towards lseq)"""
        rlen = sum(l for op, l in cigar if op in (0, 2, 3, 7, 8)) or 1
        name = qname.encode() + b"\x00"
        o = int(rng.integers(0, len(qpool) - lseq - 1))
        body = (struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, _reg2bin(pos, pos + rlen) if pos >= 0 else 4680, len(cigar), flag, lseq,
                            mtid, mpos, tlen) + name + b"".join(struct.pack("<I", (l << 4) | op) for op, l in cigar) +
                packed_seq(tid, pos, lseq, rng) + qpool[o:o + lseq] + aux)
        return struct.pack("<i", len(body)) + body

    def sa_tag(entries):
        return b"SAZ" + "".join("%s,%d,%s,%s,%d,%d;" % e for e in entries).encode() + b"\x00"

    def cig_str(cig):
        return "".join("%d%s" % (l, "MIDNSHP=X"[op]) for op, l in cig)

    M, S, H, D = 0, 4, 5, 2

    # ---------------- planted events + noise pairs (main thread, one RNG) ----------------
    events = []
    weights = lens[big] / lens[big].sum() if big else None
    n_events = int(round(sv_per_mb * lens[big].sum() / 1e6)) if big else 0

    def put(tid, pos, rec):
        specials[tid if tid >= 0 else ncon].append((pos, rec))

    def pair(qn, tA, pA, revA, cigA, tB, pB, revB, cigB, mapqA=60, mapqB=60, auxA=b"", auxB=b"", first_is_A=True, extraA=0, extraB=0):
        endA = pA + sum(l for op, l in cigA if op in (0, 2))
        endB = pB + sum(l for op, l in cigB if op in (0, 2))
        if tA == tB:
            lo, hi = min(pA, pB), max(endA, endB)
            tl = hi - lo
            tlA = tl if pA <= pB else -tl
            tlB = -tlA
        else:
            tlA = tlB = 0
        fA = 0x1 | (0x40 if first_is_A else 0x80) | (0x10 if revA else 0) | (0x20 if revB else 0) | extraA
        fB = 0x1 | (0x80 if first_is_A else 0x40) | (0x10 if revB else 0) | (0x20 if revA else 0) | extraB
        lA = sum(l for op, l in cigA if op in (0, 1, 4))
        lB = sum(l for op, l in cigB if op in (0, 1, 4))
        put(tA, pA, enc(erng, qn, fA, tA, pA, mapqA, cigA, tB, pB, tlA, auxA, lA))
        put(tB, pB, enc(erng, qn, fB, tB, pB, mapqB, cigB, tA, pA, tlB, auxB, lB))

    def supplementary(qn, tid, pos, rev, cig, mt, mp, mrev, sa, first):
        f = 0x1 | 0x800 | (0x40 if first else 0x80) | (0x10 if rev else 0) | (0x20 if mrev else 0)
        lq = sum(l for op, l in cig if op in (0, 1, 4))
        put(tid, pos, enc(erng, qn, f, tid, pos, 60, cig, mt, mp, 0, sa, lq))

    full = [(M, rl)]
    for ev in range(n_events):
        kind = ("DEL", "DEL", "DUP", "INV", "BND")[int(erng.integers(0, 5))]
        tA = big[int(erng.choice(len(big), p=weights))]
        LA = int(lens[tA])
        size = int(np.exp(erng.uniform(np.log(3 * span), np.log(40 * span))))
        a = int(erng.integers(4 * span, LA - 4 * span - size))
        b = a + size
        cA = contigs[tA][0]
        n_disc = int(erng.integers(4, 15))
        n_split = int(erng.integers(0, 9))
        low_event = erng.random() < 0.08                    # an event in a repeat: its reads map ambiguously
        def mq():
            return int(erng.integers(0, 12)) if low_event else (60 if erng.random() < 0.9 else int(erng.integers(3, 60)))
        def frag():
            return int(np.clip(erng.normal(insert, insert_sd), 2 * rl + 20, span))
        rec = {"type": kind, "chrA": cA, "posA": a, "chrB": cA, "posB": b, "n_disc": n_disc, "n_split": n_split}
        if kind == "BND":
            tB = big[int(erng.choice(len(big), p=weights))]
            while tB == tA and len(big) > 1:
                tB = big[int(erng.choice(len(big), p=weights))]
            b = int(erng.integers(4 * span, int(lens[tB]) - 4 * span))
            revB = bool(erng.integers(0, 2))
            rec.update(chrB=contigs[tB][0], posB=b)
        for k in range(n_disc + n_split):
            is_split = k >= n_disc
            ins = frag()
            qn = "%s%d_%d" % (kind.lower(), ev, k)
            first = bool(erng.integers(0, 2))
            # the split read's left part keeps j bases before the junction
            j = int(erng.integers(30, rl - 30)) if is_split else 0
            second_sa = [("chrUn_KI270302v1", int(erng.integers(1, 2000)), "+", "%dM%dS" % (rl // 2, rl - rl // 2), 0, 3)] if erng.random() < 0.2 else []
            sa_mq = 60 if erng.random() < 0.85 else int(erng.integers(0, 8))
            if kind == "DEL":
                pA = a - j if is_split else int(erng.integers(a - ins + rl, a - rl + 1))
                pB = pA + ins - rl + size
                if is_split:
                    cigA, cigS = [(M, j), (S, rl - j)], [(H, j), (M, rl - j)]
                    auxA = sa_tag([(cA, b + 1, "+", cig_str([(S, j), (M, rl - j)]), sa_mq, 0)] + second_sa)
                    supplementary(qn, tA, b, False, cigS, tA, pB, True, sa_tag([(cA, pA + 1, "+", cig_str(cigA), 60, 0)]), first)
                    both = erng.random() < 0.15           # the mate is split too (second junction-spanning read of the fragment)
                    pair(qn, tA, pA, False, cigA, tA, pB, True, full, mq(), mq(), auxA,
                         sa_tag([(cA, a - 40 + 1, "-", "%dM%dS" % (40, rl - 40), 60, 1)]) if both else b"", first)
                else:
                    pair(qn, tA, pA, False, full, tA, pB, True, full, mq(), mq(), first_is_A=first)
            elif kind == "DUP":
                pA = b - j if is_split else int(erng.integers(b - ins + rl, b - rl + 1))     # forward read at the end of the first copy
                pB = a + (pA + ins - rl - b)                                                 # its mate at the start of the second
                if pB < a:
                    pB = a
                if is_split:
                    cigA, cigS = [(M, j), (S, rl - j)], [(H, j), (M, rl - j)]
                    auxA = sa_tag([(cA, a + 1, "+", cig_str([(S, j), (M, rl - j)]), sa_mq, 0)] + second_sa)
                    supplementary(qn, tA, a, False, cigS, tA, pB, True, sa_tag([(cA, pA + 1, "+", cig_str(cigA), 60, 0)]), first)
                    pair(qn, tA, pA, False, cigA, tA, pB, True, full, mq(), mq(), auxA, b"", first)
                else:
                    pair(qn, tA, pA, False, full, tA, pB, True, full, mq(), mq(), first_is_A=first)
            elif kind == "INV":
                if k % 2 == 0:            # junction at a: forward read before a, mate inside the inverted segment reads forward
                    pA = a - j if is_split else int(erng.integers(a - ins + rl, a - rl + 1))
                    pB = max(a, b - (pA + ins - rl - a) - rl)
                    if is_split:
                        cigA = [(M, j), (S, rl - j)]
                        sp = b - (rl - j)
                        auxA = sa_tag([(cA, sp + 1, "-", cig_str([(M, rl - j), (S, j)]), sa_mq, 0)] + second_sa)
                        supplementary(qn, tA, sp, True, [(M, rl - j), (H, j)], tA, pB, False, sa_tag([(cA, pA + 1, "+", cig_str(cigA), 60, 0)]), first)
                        pair(qn, tA, pA, False, cigA, tA, pB, False, full, mq(), mq(), auxA, b"", first)
                    else:
                        pair(qn, tA, pA, False, full, tA, pB, False, full, mq(), mq(), first_is_A=first)
                else:                     # junction at b: both reads reverse
                    sA = int(erng.integers(b - ins + rl, b - rl + 1))
                    pA = min(b - rl, a + (b - sA - rl))
                    pB = sA + ins - rl
                    pair(qn, tA, pA, True, full, tA, pB, True, full, mq(), mq(), first_is_A=first)
            else:                         # BND
                pA = a - j if is_split else int(erng.integers(a - ins + rl, a - rl + 1))
                off = pA + ins - rl - a
                pB = b + off if not revB else b - off - rl
                if is_split:
                    cigA = [(M, j), (S, rl - j)]
                    if not revB:
                        sp, scig, shard = b, [(S, j), (M, rl - j)], [(H, j), (M, rl - j)]
                    else:
                        sp, scig, shard = b - (rl - j), [(M, rl - j), (S, j)], [(M, rl - j), (H, j)]
                    auxA = sa_tag([(contigs[tB][0], sp + 1, "-" if revB else "+", cig_str(scig), sa_mq, 0)] + second_sa)
                    supplementary(qn, tB, sp, revB, shard, tB, pB, not revB, sa_tag([(cA, pA + 1, "+", cig_str(cigA), 60, 0)]), first)
                    pair(qn, tA, pA, False, cigA, tB, pB, not revB, full, mq(), mq(), auxA, b"", first)
                else:
                    pair(qn, tA, pA, False, full, tB, pB, not revB, full, mq(), mq(), first_is_A=first)
        events.append(rec)
    # noise pairs: mate far away on the same contig, or on another contig
    n_noise = int(noise_pair_frac * sum(n_pairs)) if big else 0
    for k in range(n_noise):
        tA = big[int(erng.choice(len(big), p=weights))]
        pA = int(erng.integers(0, int(lens[tA]) - rl))
        tB = tA if erng.random() < 0.6 else big[int(erng.choice(len(big), p=weights))]
        pB = int(erng.integers(0, int(lens[tB]) - rl))
        pair("noise%d" % k, tA, pA, bool(erng.integers(0, 2)), full, tB, pB, bool(erng.integers(0, 2)), full,
             60 if erng.random() < 0.7 else int(erng.integers(0, 60)), 60 if erng.random() < 0.7 else int(erng.integers(0, 60)))
    for k in range(40):                   # unplaced unmapped pairs at the end of the file
        for fl in (0x1 | 0x4 | 0x8 | 0x40, 0x1 | 0x4 | 0x8 | 0x80):
            put(-1, -1, enc(erng, "unmapped%d" % k, fl, -1, -1, 0, [], -1, -1, 0))

    # ---------------- per contig: bulk pairs + per-pair variations, interleaved with the specials by position ----------------
    rec_dt = np.dtype([("block_size", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                       ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("mate_tid", "<i4"), ("mate_pos", "<i4"), ("tlen", "<i4"),
                       ("name", "S12"), ("cigar", "<u4"), ("seq", "u1", (half,)), ("qual", "u1", (rl,))])
    R = rec_dt.itemsize
    qual_lut = np.sort(np.minimum(40, 2 + (38 * np.sqrt(np.arange(256) / 255.0)).astype(np.int64)).astype(np.uint8))
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "@RG\tID:rg1\tSM:%s\n" % sample
    tb = text.encode()
    head = b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", ncon)
    for name, ln in contigs:
        nb = name.encode() + b"\x00"
        head += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)

    def names_of(ids):
        digits = (ids[:, None] // (10 ** np.arange(9, -1, -1, dtype=np.int64))[None, :] % 10 + 48).astype(np.uint8)
        return np.concatenate([np.full((len(ids), 1), ord("p"), np.uint8), digits, np.zeros((len(ids), 1), np.uint8)], axis=1).view("S12")[:, 0]

    counts = [0] * (ncon + 1)

    def contig_job(tid):
        name, L = contigs[tid]
        L = int(L)
        rng = np.random.default_rng([seed, tid])
        npair = n_pairs[tid]
        sp = specials[tid]
        part = "%s.part%03d" % (path, tid)
        if npair:
            posA = rng.integers(0, L - span, npair).astype(np.int64)
            ins = np.clip(rng.normal(insert, insert_sd, npair), rl + 1, span).astype(np.int64)
            posB = posA + ins - rl
            u = rng.random(npair)
            firstA = rng.random(npair) < 0.5
            ids = pair_base[tid] + np.arange(npair, dtype=np.int64)
            mapqA = np.where(rng.random(npair) < 0.88, 60, rng.integers(0, 60, npair)).astype(np.uint8)
            mapqB = np.where(rng.random(npair) < 0.88, 60, rng.integers(0, 60, npair)).astype(np.uint8)
            dupA = np.where(rng.random(npair) < 0.02, 0x400, 0)
            dupB = np.where(rng.random(npair) < 0.02, 0x400, 0)
            flagA = 0x1 | 0x2 | 0x20 | np.where(firstA, 0x40, 0x80) | dupA
            flagB = 0x1 | 0x2 | 0x10 | np.where(firstA, 0x80, 0x40) | dupB
            # per-pair variations: A (and for the unmapped-mate pairs B) leave the bulk arrays
            cat = np.zeros(npair, dtype=np.uint8)
            for c, p in ((1, 0.005), (2, 0.007), (3, 0.008), (4, 0.009), (5, 0.0095)):
                cat[(u < p) & (cat == 0)] = c
            nm = names_of(ids)
            for i in np.flatnonzero(cat):
                c, qn, pA, pB, tl = int(cat[i]), nm[i].decode(), int(posA[i]), int(posB[i]), int(ins[i])
                fA, fB, mA = int(flagA[i]), int(flagB[i]), int(mapqA[i])
                if c == 1:                                   # soft clip on either side
                    k = int(rng.integers(26, 61))
                    cig = [(S, k), (M, rl - k)] if rng.random() < 0.5 else [(M, rl - k), (S, k)]
                    sp.append((pA, enc(rng, qn, fA, tid, pA, mA, cig, tid, pB, tl)))
                elif c == 2:                                 # deletion inside the read
                    d = int(rng.integers(1, 31))
                    sp.append((pA, enc(rng, qn, fA, tid, pA, mA, [(M, 70), (D, d), (M, rl - 70)], tid, pB, tl)))
                elif c == 3:                                 # chimeric read: primary with SA + its supplementary alignment elsewhere
                    k = int(rng.integers(40, 80))
                    q = int(rng.integers(0, L - rl))
                    sa = sa_tag([(name, q + 1, "+", "%dS%dM" % (k, rl - k), int(rng.choice([60, 60, 60, 20, 0])), 1)])
                    sp.append((pA, enc(rng, qn, fA & ~0x2, tid, pA, mA, [(M, k), (S, rl - k)], tid, pB, tl, sa)))
                    sp.append((q, enc(rng, qn, (fA & ~0x2) | 0x800, tid, q, 60, [(H, k), (M, rl - k)], tid, pB, 0,
                                      sa_tag([(name, pA + 1, "+", "%dM%dS" % (k, rl - k), mA, 0)]), rl - k)))
                elif c == 4:                                 # secondary alignment of A somewhere else
                    q = int(rng.integers(0, L - rl))
                    sp.append((pA, enc(rng, qn, fA, tid, pA, mA, full, tid, pB, tl)))
                    sp.append((q, enc(rng, qn, fA | 0x100, tid, q, 0, full, tid, pB, 0)))
                else:                                        # mate unmapped: B sits at A's position without a CIGAR
                    fa = 0x1 | 0x8 | (fA & 0xc0)
                    sp.append((pA, enc(rng, qn, fa, tid, pA, mA, full, tid, pA, 0)))
                    sp.append((pA, enc(rng, qn, 0x1 | 0x4 | (fB & 0xc0), tid, pA, 0, [], tid, pA, 0)))
            keepA = cat == 0
            keepB = cat != 5
            pos = np.concatenate([posA[keepA], posB[keepB]])
            mate = np.concatenate([posB[keepA], posA[keepB]])
            tl = np.concatenate([ins[keepA], -ins[keepB]])
            flag = np.concatenate([flagA[keepA], flagB[keepB]])
            mapq = np.concatenate([mapqA[keepA], mapqB[keepB]])
            nm = np.concatenate([nm[keepA], nm[keepB]])
            order = np.argsort(pos, kind="stable")
            pos, mate, tl, flag, mapq, nm = pos[order], mate[order], tl[order], flag[order], mapq[order], nm[order]
        else:
            pos = np.zeros(0, dtype=np.int64)
        sp.sort(key=lambda r: r[0])                          # stable: equal positions keep generation order
        sp_pos = np.array([r[0] for r in sp], dtype=np.int64)
        sp_at = np.searchsorted(pos, sp_pos, side="right")   # special k goes in front of bulk record sp_at[k]
        counts[tid] = len(pos) + len(sp)
        ref_nib = None
        if ref_seqs is not None and len(pos):
            ref_nib = np.concatenate([_NIB[ref_seqs[name]], np.full(rl + 2, 1, np.uint8)])
        with open(part, "wb") as f:
            pend = bytearray(head if tid == 0 else b"")
            k = 0

            def flush(final=False):
                nblk = len(pend) // 0xff00 if not final else -(-len(pend) // 0xff00)
                for o in range(0, nblk * 0xff00, 0xff00):
                    f.write(_bgzf_block(bytes(pend[o:o + 0xff00]), level))
                del pend[:nblk * 0xff00]

            for lo in range(0, len(pos), chunk):
                hi = min(len(pos), lo + chunk)
                m = hi - lo
                a = np.zeros(m, dtype=rec_dt)
                a["block_size"] = R - 4
                a["tid"], a["pos"], a["l_name"] = tid, pos[lo:hi], 12
                a["mapq"], a["n_cigar"], a["l_seq"] = mapq[lo:hi], 1, rl
                a["flag"] = flag[lo:hi]
                a["mate_tid"], a["mate_pos"], a["tlen"] = tid, mate[lo:hi], tl[lo:hi]
                e1 = pos[lo:hi] + rl - 1
                b = np.full(m, 0, dtype=np.int64)
                done = np.zeros(m, dtype=bool)
                for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
                    same = ~done & ((pos[lo:hi] >> shift) == (e1 >> shift))
                    b[same] = base + (pos[lo:hi][same] >> shift)
                    done |= same
                a["bin"] = b
                a["name"] = nm[lo:hi]
                a["cigar"] = (rl << 4) | 0
                if ref_nib is not None:
                    nib = ref_nib[pos[lo:hi, None] + np.arange(rl + (rl & 1))[None, :]]
                    mism = rng.random(nib.shape) < 0.003
                    nib = np.where(mism, np.roll(nib, 1, axis=1), nib)
                else:
                    nib = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, (m, rl + (rl & 1)), dtype=np.uint8)]
                if rl & 1:
                    nib[:, rl] = 0
                a["seq"] = (nib[:, 0::2] << 4) | nib[:, 1::2]
                q = np.repeat(qual_lut[rng.integers(96, 256, (m, (rl + 9) // 10), dtype=np.uint8)], 10, axis=1)[:, :rl]
                a["qual"] = np.where(rng.random((m, rl)) < 0.1, qual_lut[rng.integers(0, 256, (m, rl), dtype=np.uint8)], q)
                buf = memoryview(a.tobytes())
                prev = lo
                while k < len(sp) and sp_at[k] < hi:
                    at = int(sp_at[k])
                    pend += buf[(prev - lo) * R:(at - lo) * R]
                    pend += sp[k][1]
                    prev = at
                    k += 1
                pend += buf[(prev - lo) * R:]
                flush()
            while k < len(sp):
                pend += sp[k][1]
                k += 1
            flush(final=True)
        return part

    with ThreadPoolExecutor(max(1, threads)) as ex:
        parts = list(ex.map(contig_job, range(ncon)))
    tail = b"".join(r[1] for r in specials[ncon])
    counts[ncon] = len(specials[ncon])
    with open(path, "wb") as out:
        for part in parts:
            with open(part, "rb") as f:
                while True:
                    blk = f.read(64 << 20)
                    if not blk:
                        break
                    out.write(blk)
            os.remove(part)
        for o in range(0, len(tail), 0xff00):
            out.write(_bgzf_block(tail[o:o + 0xff00], level))
        out.write(_BGZF_EOF)
    return {"events": events, "n_records": int(sum(counts)), "n_noise_pairs": n_noise, "n_bulk_pairs": int(sum(n_pairs))}
