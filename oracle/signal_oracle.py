"""Literal per-read restatement of tiddit_signal.pyx (worker :147-228, main :230-334) and of the
`tiddit --cov` loop (__main__.py:225-242).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the BAM-decode boundary: the reference reads alignments through pysam/htslib, which
cannot be installed here, so this restatement defines the record attributes itself (an independent
pure-Python BAM parser below; reference_end = htslib bam_endpos; query_alignment_start = leading soft
clip) and follows the reference's control flow line by line from there.  It cross-checks the product's C
decoder + vectorised predicates; it is not a substitute for a pysam-backed run.
"""
import itertools
import struct
import zlib


import oracle

_SEQ = "=ACMGRSVTWYHKDBN"


class Read:
    pass


def parse_record(raw, o, sq):
    """one alignment record at byte offset o of the inflated stream -> (Read, offset of the next record)"""
    bs = struct.unpack_from("<i", raw, o)[0]
    tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", raw, o + 4)
    p = o + 36
    r = Read()
    r.query_name = raw[p:p + l_name - 1].decode()
    p += l_name
    r.cigartuples = [(w & 0xf, w >> 4) for w in struct.unpack_from("<%dI" % n_cig, raw, p)]
    p += 4 * n_cig
    seq = []
    for i in range(l_seq):
        b = raw[p + i // 2]
        seq.append(_SEQ[(b >> 4) if i % 2 == 0 else (b & 0xf)])
    r.query_sequence = "".join(seq)
    p += (l_seq + 1) // 2 + l_seq
    r.tags = {}
    end = o + 4 + bs
    while p < end:
        tag, typ = raw[p:p + 2].decode(), chr(raw[p + 2])
        p += 3
        if typ == "Z":
            e = raw.index(b"\x00", p)
            r.tags[tag] = raw[p:e].decode()
            p = e + 1
        elif typ == "i":
            r.tags[tag] = struct.unpack_from("<i", raw, p)[0]
            p += 4
        elif typ == "A":
            r.tags[tag] = chr(raw[p])
            p += 1
        elif typ in "cCsSIf":
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "I": "I", "f": "f"}[typ]
            r.tags[tag] = struct.unpack_from("<" + fmt, raw, p)[0]
            p += struct.calcsize(fmt)
        elif typ == "H":
            e = raw.index(b"\x00", p)
            r.tags[tag] = raw[p:e].decode()
            p = e + 1
        elif typ == "B":
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[chr(raw[p])]
            cnt = struct.unpack_from("<I", raw, p + 1)[0]
            r.tags[tag] = list(struct.unpack_from("<%d%s" % (cnt, fmt), raw, p + 5))
            p += 5 + cnt * struct.calcsize(fmt)
        else:
            raise ValueError("tag type " + typ)
    r.flag, r.mapq, r.isize = flag, mapq, tlen
    r.is_unmapped, r.is_duplicate = bool(flag & 0x4), bool(flag & 0x400)
    r.is_supplementary, r.is_secondary = bool(flag & 0x800), bool(flag & 0x100)
    r.mate_is_unmapped, r.is_paired, r.is_reverse = bool(flag & 0x8), bool(flag & 0x1), bool(flag & 0x10)
    r.reference_id, r.next_reference_id = tid, mtid
    r.mate_pos = mpos
    r.reference_name = sq[tid]["SN"] if tid >= 0 else None
    r.next_reference_name = sq[mtid]["SN"] if mtid >= 0 else None
    r.reference_start = pos
    rlen = sum(l for op, l in r.cigartuples if op in (0, 2, 3, 7, 8))
    r.reference_end = pos + (rlen if (rlen and not r.is_unmapped) else 1)       # bam_endpos
    qs = 0
    for op, l in r.cigartuples:
        if op == 5:
            continue
        if op == 4:
            qs += l
        else:
            break
    r.query_alignment_start = qs
    return r, end


def parse_bam(path):
    """-> (header dict, list of Read) — independent of tiddit_amd.bamio"""
    raw = bytearray()
    with open(path, "rb") as f:
        data = f.read()
    o = 0
    while o < len(data):
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        bsize = struct.unpack_from("<H", data, o + 16)[0]
        cdata = data[o + 12 + xlen:o + bsize + 1 - 8]
        raw += zlib.decompress(bytes(cdata), -15) if cdata else b""
        o += bsize + 1
    raw = bytes(raw)
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].split(b"\x00")[0].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    sq = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        name = raw[o + 4:o + 4 + ln - 1].decode()
        sq.append({"SN": name, "LN": struct.unpack_from("<i", raw, o + 4 + ln)[0]})
        o += 8 + ln
    header = {"SQ": sq}
    for line in text.split("\n"):
        if line.startswith("@RG"):
            header.setdefault("RG", []).append(dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f))
    reads = []
    while o < len(raw):
        r, o = parse_record(raw, o, sq)
        reads.append(r)
    return header, reads


def cov_main(header, reads, z, q):
    """__main__.py:225-242 -> dict contig -> float64 bins"""
    cov, ebs = {}, {}
    for c in header["SQ"]:
        cov[c["SN"]], ebs[c["SN"]] = oracle.create_coverage(c["LN"], z)
    for read in reads:
        if read.is_unmapped or read.is_duplicate:
            continue
        if read.mapq >= q:
            oracle.update_coverage(read.reference_start, read.reference_end, z, cov[read.reference_name], ebs[read.reference_name])
    return cov


def _find_SA_query_range(SA):            # tiddit_signal.pyx:11-29
    SC = ["".join(x) for _, x in itertools.groupby(SA[3], key=str.isdigit)]
    s_to_op = {"M": 0, "S": 4, "H": 5, "D": 2, "I": 1}
    cig = [(s_to_op[SC[i * 2 + 1]], int(SC[i * 2])) for i in range(0, int(len(SC) / 2))]
    a = Read()
    a.reference_start = int(SA[1])
    ref = sum(l for op, l in cig if op in (0, 2))
    a.reference_end = a.reference_start + (ref if ref else 1)
    qs = 0
    for op, l in cig:
        if op == 5:
            continue
        if op == 4:
            qs += l
        else:
            break
    a.query_alignment_start = qs
    a.query_alignment_end = qs + sum(l for op, l in cig if op in (0, 1))
    return a


def _SA_analysis(read, min_q, reference_name):      # tiddit_signal.pyx:31-145, statement by statement
    sas = read.tags["SA"].rstrip(";").split(";")
    if len(sas) > 1:
        SA_lengths, ok_q = [], []
        for i in range(0, len(sas)):
            SA_data = sas[0].split(",")
            if int(SA_data[4]) >= min_q:
                ok_q.append(i)
                s = _find_SA_query_range(SA_data)
                SA_lengths.append(s.query_alignment_end - s.query_alignment_start)
        longest = 0
        for i in range(0, len(ok_q)):
            if SA_lengths[i] > SA_lengths[longest]:
                longest = i
        if len(ok_q) == 0:
            return ()
        elif len(ok_q) == 1:
            sas[0] = sas[ok_q[0]]
        else:
            sas[0] = sas[longest]
    SA_data = sas[0].split(",")
    if int(SA_data[4]) < min_q:
        return ()
    clip_before = False
    s = _find_SA_query_range(SA_data)
    if s.query_alignment_start < read.query_alignment_start:
        clip_before = True
    if not clip_before:
        split_pos = read.reference_start + 1 if read.is_reverse else read.reference_end + 1
    else:
        split_pos = read.reference_end + 1 if read.is_reverse else read.reference_start + 1
    SA_chr = SA_data[0]
    startA, endA = read.reference_start + 1, read.reference_end + 1
    startB, endB = s.reference_start, s.reference_end
    if clip_before:
        SA_split_pos = s.reference_start if SA_data[2] == "-" else s.reference_end
    else:
        SA_split_pos = s.reference_end if SA_data[2] == "-" else s.reference_start
    if SA_chr < reference_name:
        chrA, chrB = SA_chr, reference_name
        split_pos, SA_split_pos = SA_split_pos, split_pos
        startB, endB = read.reference_start + 1, read.reference_end + 1
        startA, endA = s.reference_start, s.reference_end
    else:
        chrA, chrB = reference_name, SA_chr
        if chrA == chrB:
            if SA_split_pos < split_pos:
                split_pos, SA_split_pos = SA_split_pos, split_pos
                startB, endB = read.reference_start + 1, read.reference_end + 1
                startA, endA = s.reference_start, s.reference_end
    return [chrA, chrB, read.query_name, split_pos, read.is_reverse, SA_split_pos, "-" == SA_data[2], startA, endA, startB, endB]


def _merge_and_format(header, chromosomes, res):
    """tiddit_signal.main :262-326 — merge the workers' rows and format discordants.tab / splits.tab"""
    data = {a: {b["SN"]: {} for b in header["SQ"]} for a in chromosomes}
    splits = {a: {b["SN"]: {} for b in header["SQ"]} for a in chromosomes}
    coverage_data = {}
    for chromosome, d, sp, cov in res:
        coverage_data[chromosome] = cov
        for signal in d:
            if signal[0] not in data:
                continue
            data[signal[0]][signal[1]].setdefault(signal[2], []).append(signal[3:])
        for signal in sp:
            if signal[0] not in splits:
                continue
            splits[signal[0]][signal[1]].setdefault(signal[2], [])
            splits[signal[0]][signal[1]][signal[2]] += signal[3:]
    disc_txt = []
    for chrA in data:
        for chrB in data[chrA]:
            for fragment in data[chrA][chrB]:
                fr = data[chrA][chrB][fragment]
                if len(fr) < 2:
                    continue
                if chrA == chrB:
                    if fr[1][-1] < fr[0][-1]:
                        out = fr[1][0:-1] + fr[0][0:-1]
                    else:
                        out = fr[0][0:-1] + fr[1][0:-1]
                else:
                    if fr[0][-1] == chrA:
                        out = fr[0][0:-1] + fr[1][0:-1]
                    else:
                        out = fr[1][0:-1] + fr[0][0:-1]
                disc_txt.append("{}\t{}\t{}\t{}\n".format(fragment, chrA, chrB, "\t".join(map(str, out))))
    split_txt = []
    for chrA in splits:
        for chrB in splits[chrA]:
            for fragment in splits[chrA][chrB]:
                split_txt.append("{}\t{}\t{}\t{}\n".format(fragment, chrA, chrB, "\t".join(map(str, splits[chrA][chrB][fragment]))))
    return coverage_data, "".join(disc_txt), "".join(split_txt)


def signal_main(header, reads, min_q, max_ins, sample_id, min_contig, min_anchor_len, min_clip_len):
    """-> (coverage dict, discordants.tab text, splits.tab text, clips.fa text, per-contig clip texts)"""
    bin_size = 50
    chromosomes = [c["SN"] for c in header["SQ"] if c["LN"] >= min_contig]
    data = {a: {b["SN"]: {} for b in header["SQ"]} for a in chromosomes}
    splits = {a: {b["SN"]: {} for b in header["SQ"]} for a in chromosomes}
    coverage_data, clip_texts = {}, {}
    res = []
    for chromosome in chromosomes:           # worker(), one contig at a time
        LN = [c["LN"] for c in header["SQ"] if c["SN"] == chromosome][0]
        cov, ebs = oracle.create_coverage(LN, bin_size)
        clips, d, sp = [], [], []
        for read in reads:
            if read.reference_name != chromosome:
                continue
            if read.is_unmapped or read.is_duplicate:
                continue
            read_chromosome, mate_chromosome = read.reference_name, read.next_reference_name
            if read.mapq >= min_q:
                oracle.update_coverage(read.reference_start, read.reference_end, bin_size, cov, ebs)
            if read.is_supplementary or read.is_secondary:
                continue
            if read.mapq < min_q:
                continue
            if abs(read.isize) < max_ins and mate_chromosome == read_chromosome:
                ct = read.cigartuples
                if (ct[0][0] == 4 and ct[0][1] > min_clip_len) and (ct[-1][0] == 0 and ct[-1][1] > min_anchor_len):
                    clips.append([">{}|{}|{}\n".format(read.query_name, read_chromosome, read.reference_start + 1), read.query_sequence + "\n"])
                elif ct[-1][0] == 4 and ct[-1][1] > min_clip_len and (ct[0][0] == 0 and ct[0][1] > min_anchor_len):
                    clips.append([">{}|{}|{}\n".format(read.query_name, read_chromosome, read.reference_start + 1), read.query_sequence + "\n"])
            if "SA" in read.tags:
                split = _SA_analysis(read, min_q, read_chromosome)
                if split:
                    sp.append(split)
            if read.mate_is_unmapped:
                continue
            if not read.is_paired:
                continue
            if abs(read.isize) > max_ins or mate_chromosome != read_chromosome:
                if mate_chromosome < read_chromosome:
                    chrA, chrB = mate_chromosome, read_chromosome
                else:
                    chrA, chrB = read_chromosome, mate_chromosome
                d.append([chrA, chrB, read.query_name, read.reference_start + 1, read.reference_end + 1, read.is_reverse, read_chromosome])
        clip_texts[chromosome] = "".join("".join(c) for c in clips)
        res.append((chromosome, d, sp, cov))
    coverage_data, disc, split = _merge_and_format(header, chromosomes, res)
    return coverage_data, disc, split, "".join(clip_texts[c] for c in chromosomes), clip_texts


def inflate_bam(path):
    """-> (header dict, sq list, uint8 array of the inflated record bytes) — zlib block by block, independent of tiddit_amd.bamio"""
    import numpy as np
    parts = []
    with open(path, "rb") as f:
        data = f.read()
    o = 0
    while o < len(data):
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        bsize = struct.unpack_from("<H", data, o + 16)[0]
        cdata = data[o + 12 + xlen:o + bsize + 1 - 8]
        if cdata:
            parts.append(zlib.decompress(cdata, -15))
        o += bsize + 1
    del data
    raw = b"".join(parts)
    del parts
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].split(b"\x00")[0].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    sq = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        sq.append({"SN": raw[o + 4:o + 4 + ln - 1].decode(), "LN": struct.unpack_from("<i", raw, o + 4 + ln)[0]})
        o += 8 + ln
    header = {"SQ": sq}
    for line in text.split("\n"):
        if line.startswith("@RG"):
            header.setdefault("RG", []).append(dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f))
    return header, sq, np.frombuffer(raw, dtype=np.uint8)[o:]


def signal_main_file(path, min_q, max_ins, sample_id, min_contig, min_anchor_len, min_clip_len, want_clips=True):
    """tiddit_signal.main (:230-334) on a BAM file at scale: the per-read chain of worker (:169-221) runs in C over the decoded
    fields (oracle.bam_walk + oracle.signal_worker), and only the reads it marks are parsed record by record and pushed through
    the literal Python of the rest (clip entries, _SA_analysis, discordant rows, the merge and the writers — shared with
    signal_main).  -> (coverage dict, discordants.tab text, splits.tab text, clips.fa text, per-contig clip texts, n_reads)"""
    import numpy as np
    max_ins = int(max_ins)                 # `int max_ins` (:230)
    header, sq, raw = inflate_bam(path)
    f = oracle.bam_walk(raw)
    chromosomes = [c["SN"] for c in sq if c["LN"] >= min_contig]
    res, clip_texts = _workers_over_fields(sq, raw, f, chromosomes, min_q, max_ins, min_anchor_len, min_clip_len, want_clips)
    cov, disc, split = _merge_and_format(header, chromosomes, res)
    return cov, disc, split, "".join(clip_texts[c] for c in chromosomes), clip_texts, len(f["tid"])


def _workers_over_fields(sq, raw, f, chromosomes, min_q, max_ins, min_anchor_len, min_clip_len, want_clips=True):
    """tiddit_signal.worker (:147-228) for every contig of `chromosomes` over the decoded fields `f` of the inflated records `raw`
    -> ([(chromosome, discordant rows, split rows, coverage)], {chromosome: clip FASTA text})"""
    import numpy as np
    tid = f["tid"]
    names = [c["SN"] for c in sq]
    res, clip_texts = [], {}
    # samfile.fetch(chromosome): the contig's records, in file order (a coordinate-sorted file keeps them together; an
    # unsorted one is gathered by a stable selection)
    order = np.argsort(tid, kind="stable")
    bounds = np.searchsorted(tid[order], np.arange(len(sq) + 1))
    contiguous = bool(np.all(np.diff(tid[tid >= 0]) >= 0)) if len(tid) else True
    for chromosome in chromosomes:
        t = names.index(chromosome)
        lo, hi = int(bounds[t]), int(bounds[t + 1])
        if contiguous:
            first = int(order[lo]) if hi > lo else 0
            g = f
            sel_lo, sel_hi = first, first + (hi - lo)
        else:
            idx = order[lo:hi]
            g = {k: v[idx] for k, v in f.items()}
            sel_lo, sel_hi = 0, hi - lo
        act, cov = oracle.signal_worker(g, sel_lo, sel_hi, sq[t]["LN"], min_q, max_ins, min_anchor_len, min_clip_len, 50)
        clips, d, sp = [], [], []
        for k in np.flatnonzero(act & 0xe):
            a = int(act[k])
            off = int(g["rec_off"][sel_lo + k])
            bs = int(raw[off:off + 4].view("<i4")[0])
            read, _ = parse_record(raw[off:off + 4 + bs].tobytes(), 0, sq)
            read_chromosome, mate_chromosome = read.reference_name, read.next_reference_name
            if a & 2 and want_clips:
                clips.append([">{}|{}|{}\n".format(read.query_name, read_chromosome, read.reference_start + 1), read.query_sequence + "\n"])
            if a & 4:
                split = _SA_analysis(read, min_q, read_chromosome)
                if split:
                    sp.append(split)
            if a & 8:
                if mate_chromosome < read_chromosome:
                    chrA, chrB = mate_chromosome, read_chromosome
                else:
                    chrA, chrB = read_chromosome, mate_chromosome
                d.append([chrA, chrB, read.query_name, read.reference_start + 1, read.reference_end + 1, read.is_reverse, read_chromosome])
        clip_texts[chromosome] = "".join("".join(c) for c in clips)
        res.append((chromosome, d, sp, cov))
    return res, clip_texts


# ---------------------------------------------------------------------------------------------------------------------
# A BOUNDED SAMPLE of a large coordinate-sorted BAM without an index: the records of a run of contigs, found by a binary search
# over the BGZF blocks (each block inflates on its own; a record start inside it is recognised by its fields and confirmed by the
# chain of block_size hops that follows).  bench.py times the restatement on such a sample when the whole file would take minutes
# (a 3-Gb genome: 54 GB of BAM, ~9 minutes on one core), and compares the product's outputs for those contigs with it.

def read_header(path):
    """-> (header dict, sq list): inflates only the blocks that hold the header"""
    raw = b""
    with open(path, "rb") as f:
        while True:
            head = f.read(18)
            if len(head) < 18:
                break
            xlen, bsize = struct.unpack_from("<H", head, 10)[0], struct.unpack_from("<H", head, 16)[0]
            rest = f.read(bsize + 1 - 18)
            cdata = (head + rest)[12 + xlen:bsize + 1 - 8]
            raw += zlib.decompress(cdata, -15) if cdata else b""
            if len(raw) >= 12:
                l_text = struct.unpack_from("<i", raw, 4)[0]
                if len(raw) >= 12 + l_text:
                    n_ref = struct.unpack_from("<i", raw, 8 + l_text)[0]
                    o, ok = 12 + l_text, True
                    for _ in range(n_ref):
                        if len(raw) < o + 4:
                            ok = False
                            break
                        ln = struct.unpack_from("<i", raw, o)[0]
                        o += 8 + ln
                    if ok and len(raw) >= o:
                        break
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].split(b"\x00")[0].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    sq = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        sq.append({"SN": raw[o + 4:o + 4 + ln - 1].decode(), "LN": struct.unpack_from("<i", raw, o + 4 + ln)[0]})
        o += 8 + ln
    header = {"SQ": sq}
    for line in text.split("\n"):
        if line.startswith("@RG"):
            header.setdefault("RG", []).append(dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f))
    return header, sq


def block_offsets(path):
    """file offsets of every BGZF block (header hops over a memory map; nothing is inflated)"""
    import mmap
    import numpy as np
    offs = []
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        o, n = 0, len(mm)
        while o + 18 <= n:
            offs.append(o)
            o += struct.unpack_from("<H", mm, o + 16)[0] + 1
        mm.close()
    return np.array(offs + [o], dtype=np.int64)


def _inflate_blocks(mm, offs, b0, b1):
    parts = []
    for b in range(b0, b1):
        o, e = int(offs[b]), int(offs[b + 1])
        xlen = struct.unpack_from("<H", mm, o + 10)[0]
        cdata = mm[o + 12 + xlen:e - 8]
        if cdata:
            parts.append(zlib.decompress(cdata, -15))
    return b"".join(parts)


def _record_at(buf, o, sq):
    """does a plausible BAM record start at buf[o]?  (SAM/BAM spec v1 §4.2 field ranges against the header)"""
    if o + 36 > len(buf):
        return False
    bs, tid, pos = struct.unpack_from("<iii", buf, o)
    l_name, n_cig = buf[o + 12], struct.unpack_from("<H", buf, o + 16)[0]
    l_seq, mtid, mpos = struct.unpack_from("<iii", buf, o + 20)
    if bs < 32 + l_name or bs > (1 << 24) or l_name < 2 or l_seq < 0 or not (-1 <= tid < len(sq)) or not (-1 <= mtid < len(sq)):
        return False
    if tid >= 0 and not (-1 <= pos <= sq[tid]["LN"]):
        return False
    if mtid >= 0 and not (-1 <= mpos <= sq[mtid]["LN"]):
        return False
    return 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq <= bs


def sync_block(mm, offs, b, sq, span=4, chain=16):
    """-> (offset of the first record that starts in block b, relative to the block's first inflated byte; its tid; the inflated
    bytes of blocks b .. b+span).  A candidate offset counts when `chain` records follow it hop by hop, all plausible."""
    nb = len(offs) - 1
    buf = _inflate_blocks(mm, offs, b, min(nb, b + span))
    first_len = len(_inflate_blocks(mm, offs, b, b + 1))
    for o in range(0, first_len):
        if not _record_at(buf, o, sq):
            continue
        p, ok, k = o, True, 0
        while k < chain and p + 36 <= len(buf):
            if not _record_at(buf, p, sq):
                ok = False
                break
            p += 4 + struct.unpack_from("<i", buf, p)[0]
            k += 1
        if ok and (k == chain or p >= len(buf) - 36):
            return o, struct.unpack_from("<i", buf, o + 4)[0], buf
    return None, None, buf


def contig_block(mm, offs, sq, tid, first_data_block):
    """the last block whose first record belongs to a contig before `tid` (the first record of `tid` starts in it or right behind it)"""
    n_ref = len(sq)
    key = lambda t: n_ref if t < 0 else t                 # the unplaced tail sorts last
    lo, hi = first_data_block, len(offs) - 1              # invariant: key(first record of block lo) < tid  (or lo is the first data block)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        o, t, _ = sync_block(mm, offs, mid, sq)
        probe = mid
        while o is None and probe + 1 < hi:              # a block without a record start (one long record): look further right
            probe += 1
            o, t, _ = sync_block(mm, offs, probe, sq)
        if o is None or key(t) >= tid:
            hi = mid
        else:
            lo = mid
    return lo


def inflate_contigs(path, sq, tid_first, tid_last, offs=None):
    """-> uint8 array of the inflated records of contigs tid_first .. tid_last (starts at a record; may carry a few records of the
    neighbouring contigs at either end — the per-contig workers select by tid)"""
    import mmap
    import numpy as np
    offs = block_offsets(path) if offs is None else offs
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        # the first block that holds records: the header ends somewhere inside block h
        first_data = 0
        b0 = contig_block(mm, offs, sq, tid_first, first_data) if tid_first > 0 else 0
        b1 = contig_block(mm, offs, sq, tid_last + 1, b0) + 2 if tid_last + 1 < len(sq) else len(offs) - 1
        b1 = min(b1 + 2, len(offs) - 1)
        if b0 == 0:                                        # from the header on: skip it
            buf = _inflate_blocks(mm, offs, 0, b1)
            l_text = struct.unpack_from("<i", buf, 4)[0]
            o = 12 + l_text
            for _ in range(struct.unpack_from("<i", buf, 8 + l_text)[0]):
                o += 8 + struct.unpack_from("<i", buf, o)[0]
        else:
            o, _, _ = sync_block(mm, offs, b0, sq)
            while o is None:
                b0 += 1
                o, _, _ = sync_block(mm, offs, b0, sq)
            buf = _inflate_blocks(mm, offs, b0, b1)
        mm.close()
    return np.frombuffer(buf, dtype=np.uint8)[o:]


def _contig_job(args):
    path, sq, t, min_q, max_ins, min_anchor_len, min_clip_len, offs = args
    raw = inflate_contigs(path, sq, t, t, offs)
    f = oracle.bam_walk(raw)
    res, clips = _workers_over_fields(sq, raw, f, [sq[t]["SN"]], min_q, max_ins, min_anchor_len, min_clip_len)
    return res[0], clips[sq[t]["SN"]], int((f["tid"] == t).sum())


def signal_main_sample(path, tids, min_q, max_ins, min_anchor_len, min_clip_len, processes=1):
    """tiddit_signal.main restricted to the contigs `tids` (a run of consecutive contig ids) of a large coordinate-sorted BAM:
    -> (coverage dict, discordants.tab text, splits.tab text, per-contig clip texts, records of those contigs).  Rows whose two contigs
    both lie in the run are exactly the full run's rows for those contig pairs (a discordant row needs both reads; split rows with the
    other contig outside the run are incomplete and must be dropped by the caller).
    processes > 1: one worker process per contig (the reference's own fan-out, tiddit_signal.pyx:259)."""
    max_ins = int(max_ins)
    header, sq = read_header(path)
    offs = block_offsets(path)
    chromosomes = [sq[t]["SN"] for t in tids]
    jobs = [(path, sq, t, min_q, max_ins, min_anchor_len, min_clip_len, offs) for t in tids]
    if processes > 1:
        import multiprocessing
        with multiprocessing.get_context("fork").Pool(min(processes, len(jobs))) as pool:
            out = pool.map(_contig_job, jobs)
    else:
        out = [_contig_job(j) for j in jobs]
    res = [o[0] for o in out]
    clip_texts = {c: o[1] for c, o in zip(chromosomes, out)}
    cov, disc, split = _merge_and_format(header, chromosomes, res)
    return cov, disc, split, clip_texts, sum(o[2] for o in out)


def statistics_prefix(path, min_mapq, max_ins_len, n_reads):
    """tiddit_stats.statistics (tiddit_stats.py:5-78) on the first `n_reads` placed alignments of a coordinate-sorted BAM: only the
    blocks that hold them are inflated (zlib), the sampling loop (:17-47) is evaluated on the decoded columns, and the figures come
    from the same numpy calls (:52-56).  -> library dict (no printing).  Pinned by the `library` entry of tests/golden/sv_e2e*.json,
    which the reference's own tiddit_stats.py produced."""
    import mmap
    import numpy as np
    header, sq = read_header(path)
    offs = block_offsets(path)
    lens, ins = [], []
    innie = outtie = 0
    sampled = 0
    done = False
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        b, carry, first = 0, b"", True
        step = 2048                                                  # blocks per piece (~130 MB inflated)
        while b < len(offs) - 1 and not done:
            b1 = min(len(offs) - 1, b + step)
            buf = carry + _inflate_blocks(mm, offs, b, b1)
            b = b1
            o = 0
            if first:
                l_text = struct.unpack_from("<i", buf, 4)[0]
                o = 12 + l_text
                for _ in range(struct.unpack_from("<i", buf, 8 + l_text)[0]):
                    o += 8 + struct.unpack_from("<i", buf, o)[0]
                first = False
            raw = np.frombuffer(buf, dtype=np.uint8)[o:]
            f = oracle.bam_walk(raw)
            k = len(f["tid"])
            used = int(f["rec_off"][k - 1]) + 4 + int(raw[int(f["rec_off"][k - 1]):int(f["rec_off"][k - 1]) + 4].view("<i4")[0]) if k else 0
            carry = bytes(raw[used:])
            placed = f["tid"] >= 0                                   # samfile.fetch() leaves the unplaced tail out (:17)
            idx = sampled + np.cumsum(placed)
            in_len = placed & (idx <= n_reads + 1)                   # read_length.append comes before the n_sampled test (:19-23)
            lens.append(f["l_seq"][in_len].astype(np.int64))
            act = placed & (idx <= n_reads)
            fl = f["flag"].astype(np.int64)
            rev, mrev = (fl & 0x10) != 0, (fl & 0x20) != 0
            ok = act & ((fl & 0x8) == 0) & (rev != mrev) & (f["mate_tid"] == f["tid"]) & (f["tlen"].astype(np.int64) <= max_ins_len) \
                & (f["mate_pos"] >= f["pos"]) & ((fl & 0xd00) == 0) & (f["mapq"].astype(np.int64) >= min_mapq)      # :25-38
            ins.append(f["tlen"][ok].astype(np.int64))
            outtie += int((ok & rev & ~mrev).sum())
            innie += int((ok & ~(rev & ~mrev)).sum())
            sampled = int(idx[-1]) if k else sampled
            done = sampled > n_reads
        mm.close()
    read_length = np.concatenate(lens) if lens else np.zeros(0, np.int64)
    insert_size = np.concatenate(ins) if ins else np.zeros(0, np.int64)
    library = {"avg_read_length": np.average(read_length)}
    if len(insert_size):
        library["avg_insert_size"] = np.average(insert_size)
        library["std_insert_size"] = np.std(insert_size)
        library["percentile_insert_size"] = np.percentile(insert_size, 99.9)
    else:
        library["avg_insert_size"] = library["std_insert_size"] = library["percentile_insert_size"] = 0
    library["mp"] = not (innie > outtie)
    return library
