// Host-side text writer for the coverage tables (no device code): the rows of print_coverage (tiddit_coverage.pyx:22-45).
// The reference formats every bin with `"{}".format(numpy.float64)` in a Python loop — 6 M rows for a human genome at
// 500-bp bins.  Here the rows are produced by the host thread pool: shortest round-trip digits from std::to_chars, laid out
// by Python's repr rules (fixed notation for 1e-4 <= |x| < 1e16 with a trailing ".0" on integers, otherwise d.ddde[+-]XX
// with at least two exponent digits), so the file is byte-identical to the reference's.
#include "tdt_common.h"

#include <charconv>
#include <string>
#include <thread>

// repr(float) of a finite or special double into p; returns the number of bytes written (<= 32)
static size_t fmt_repr(double v, char *p) {
    char *const p0 = p;
    if (v != v) {
        memcpy(p, "nan", 3);
        return 3;
    }
    if (v < 0 || (v == 0 && 1.0 / v < 0)) {
        *p++ = '-';
        v = -v;
    }
    if (v > 1.7976931348623157e308) {
        memcpy(p, "inf", 3);
        return (size_t)(p + 3 - p0);
    }
    if (v == 0) {
        memcpy(p, "0.0", 3);
        return (size_t)(p + 3 - p0);
    }
    char s[40];
    const auto r = std::to_chars(s, s + sizeof s, v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest digits
    char digits[24];
    int nd = 0;
    const char *q = s;
    for (; q < r.ptr && *q != 'e'; q++)
        if (*q != '.') digits[nd++] = *q;
    int ex = 0;
    {
        const char *e = q + 1;
        const bool neg = *e == '-';
        e++;
        for (; e < r.ptr; e++) ex = ex * 10 + (*e - '0');
        if (neg) ex = -ex;
    }
    const int decpt = ex + 1;                                   // value = 0.d1d2... * 10^decpt
    if (decpt <= -4 || decpt > 16) {                            // exponent notation
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            memcpy(p, digits + 1, (size_t)nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int x = decpt - 1;
        *p++ = x < 0 ? '-' : '+';
        if (x < 0) x = -x;
        if (x >= 100) *p++ = (char)('0' + x / 100);
        *p++ = (char)('0' + (x / 10) % 10);
        *p++ = (char)('0' + x % 10);
    } else if (decpt <= 0) {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; i++) *p++ = '0';
        memcpy(p, digits, (size_t)nd);
        p += nd;
    } else if (decpt >= nd) {
        memcpy(p, digits, (size_t)nd);
        p += nd;
        for (int i = nd; i < decpt; i++) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
    } else {
        memcpy(p, digits, (size_t)decpt);
        p += decpt;
        *p++ = '.';
        memcpy(p, digits + decpt, (size_t)(nd - decpt));
        p += nd - decpt;
    }
    return (size_t)(p - p0);
}

static inline size_t fmt_u64(unsigned long long v, char *p) {
    char t[24];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (int i = 0; i < n; i++) p[i] = t[n - 1 - i];
    return (size_t)n;
}

// One contig's rows.  kind 0 = bed (`name \t 1+i*bin \t (i+1)*bin+1 \t value`, the last row ends at contig_len,
// tiddit_coverage.pyx:34-41), kind 1 = wig (one value per line, :42-44).  Appends to `out`.
static void fmt_rows(const double *v, size_t lo, size_t hi, size_t n, const char *name, size_t name_len, long long bin, long long contig_len,
                     int kind, std::string &out) {
    out.reserve(out.size() + (hi - lo) * (kind ? 22 : 48 + name_len));
    char row[160];
    for (size_t i = lo; i < hi; i++) {
        char *p = row;
        if (kind == 0) {
            memcpy(p, name, name_len);
            p += name_len;
            *p++ = '\t';
            p += fmt_u64((unsigned long long)(1 + (long long)i * bin), p);
            *p++ = '\t';
            const long long end = i == n - 1 ? contig_len : ((long long)i + 1) * bin + 1;
            if (end < 0) {
                *p++ = '-';
                p += fmt_u64((unsigned long long)(-end), p);
            } else p += fmt_u64((unsigned long long)end, p);
            *p++ = '\t';
        }
        p += fmt_repr(v[i], p);
        *p++ = '\n';
        out.append(row, (size_t)(p - row));
    }
}

extern "C" int tdt_format_coverage(const double *values, size_t n, const char *name, int64_t bin_size, int64_t contig_len, int kind,
                                   char *out, size_t out_cap, size_t *out_len) {
    if ((!values && n) || !name || !out_len || (kind != 0 && kind != 1) || strlen(name) > 100) {
        tdt_set_error("tdt_format_coverage: bad argument");
        return TDT_E_ARG;
    }
    const size_t name_len = strlen(name);
    int threads = tdt_host_thread_count();
    if (n < 65536) threads = 1;
    std::vector<std::string> parts((size_t)threads);
    const size_t per = (n + threads - 1) / threads;
    auto work = [&](int t) {
        const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) fmt_rows(values, lo, hi, n, name, name_len, bin_size, contig_len, kind, parts[(size_t)t]);
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto &t : pool) t.join();
    }
    size_t total = 0;
    for (auto &s : parts) total += s.size();
    *out_len = total;
    if (!out || total > out_cap) return out ? TDT_E_RANGE : TDT_OK;   // out == NULL: size query
    size_t o = 0;
    for (auto &s : parts) {
        memcpy(out + o, s.data(), s.size());
        o += s.size();
    }
    return TDT_OK;
}

// ---- FASTA index (.fai) -----------------------------------------------------------------------------------------------
// What pysam.faidx(ref) produces when the index is missing (__main__.py:95-97): per sequence its name (up to the first white
// space), number of bases, byte offset of the first base, bases per line and bytes per line (from the first sequence line).
// One buffered pass with memchr instead of a Python loop over ~50 M lines for a human genome.
extern "C" int tdt_fasta_write_fai(const char *fasta_path, const char *fai_path) {
    if (!fasta_path || !fai_path) {
        tdt_set_error("tdt_fasta_write_fai: bad argument");
        return TDT_E_ARG;
    }
    FILE *f = fopen(fasta_path, "rb");
    if (!f) {
        tdt_set_error("tdt_fasta_write_fai: cannot open %s", fasta_path);
        return TDT_E_ARG;
    }
    std::string out;
    std::vector<char> buf(8u << 20);
    std::string line;                       // the current line's bytes when it straddles two reads (headers only need this)
    std::string name;
    bool have = false;
    long long length = 0, offset = 0, linebases = 0, linewidth = 0, pos = 0, cur_len = 0;
    bool cur_is_header = false, at_line_start = true;
    long long cur_bases = 0;                // bases of the current line so far (bytes that are not CR / LF)
    auto flush_entry = [&]() {
        if (have) {
            char t[128];
            snprintf(t, sizeof t, "\t%lld\t%lld\t%lld\t%lld\n", length, offset, linebases, linewidth);
            out += name;
            out += t;
        }
    };
    auto end_line = [&](bool had_eol) {     // the current line is complete (cur_len bytes including its end-of-line bytes)
        if (cur_is_header) {
            flush_entry();
            size_t e = 1;
            while (e < line.size() && line[e] != ' ' && line[e] != '\t' && line[e] != '\r' && line[e] != '\n') e++;
            name.assign(line, 1, e - 1);
            have = true;
            length = linebases = linewidth = 0;
            offset = pos;                    // pos = offset of the byte after this line
        } else if (have) {
            if (linebases == 0 && cur_bases) {
                linebases = cur_bases;
                linewidth = cur_len;
            }
            length += cur_bases;
        }
        (void)had_eol;
        line.clear();
        cur_len = cur_bases = 0;
        cur_is_header = false;
        at_line_start = true;
    };
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        size_t i = 0;
        while (i < got) {
            if (at_line_start) {
                cur_is_header = buf[i] == '>';
                at_line_start = false;
            }
            const char *nl = (const char *)memchr(buf.data() + i, '\n', got - i);
            const size_t j = nl ? (size_t)(nl - buf.data()) + 1 : got;   // end (exclusive) of this line's bytes in the buffer
            if (cur_is_header) line.append(buf.data() + i, j - i);
            else {
                long long b = (long long)(j - i);
                if (nl) b--;                                              // the LF
                if (j - i >= (nl ? 2u : 1u) && buf[j - (nl ? 2 : 1)] == '\r') b--;   // a CR in front of it (or at the buffer edge)
                cur_bases += b;
            }
            cur_len += (long long)(j - i);
            pos += (long long)(j - i);
            i = j;
            if (nl) end_line(true);
        }
    }
    if (!at_line_start) end_line(false);
    flush_entry();
    fclose(f);
    FILE *o = fopen(fai_path, "wb");
    if (!o || fwrite(out.data(), 1, out.size(), o) != out.size()) {
        if (o) fclose(o);
        tdt_set_error("tdt_fasta_write_fai: cannot write %s", fai_path);
        return TDT_E_ARG;
    }
    fclose(o);
    return TDT_OK;
}

// ---- clip FASTA entries (tiddit_signal.pyx:192-197): ">{query_name}|{contig}|{pos+1}\n{query_sequence}\n" for selected records
extern "C" int tdt_format_clips(const void *meta_, const uint32_t *raw_end, const uint8_t *raw, const uint32_t *which, size_t m, const char *contig,
                                char *out, size_t out_cap, size_t *out_len) {
    if (!out_len || (m && (!meta_ || !raw_end || !raw || !which || !contig))) {
        tdt_set_error("tdt_format_clips: bad argument");
        return TDT_E_ARG;
    }
    static const char SEQ[] = "=ACMGRSVTWYHKDBN";
    const uint8_t *meta = (const uint8_t *)meta_;
    const size_t clen = strlen(contig);
    size_t need = 0;
    for (size_t k = 0; k < m; k++) {
        const uint32_t r = which[k];
        const uint8_t *rec = raw + (r ? raw_end[r - 1] : 0u) + 4;
        int32_t lseq;
        memcpy(&lseq, rec + 16, 4);
        need += 1 + (size_t)(rec[8] ? rec[8] - 1 : 0) + 1 + clen + 1 + 11 + 1 + (size_t)(lseq > 0 ? lseq : 0) + 1;
    }
    if (!out) {
        *out_len = need;
        return TDT_OK;
    }
    if (out_cap < need) {
        tdt_set_error("tdt_format_clips: buffer of %zu bytes, %zu needed", out_cap, need);
        return TDT_E_ARG;
    }
    char *p = out;
    for (size_t k = 0; k < m; k++) {
        const uint32_t r = which[k];
        const uint8_t *rec = raw + (r ? raw_end[r - 1] : 0u) + 4;
        const int l_name = rec[8];
        uint16_t n_cig;
        int32_t lseq, pos;
        memcpy(&n_cig, rec + 12, 2);
        memcpy(&lseq, rec + 16, 4);
        memcpy(&pos, meta + (size_t)r * 28 + 8, 4);
        *p++ = '>';
        const int nl = l_name ? l_name - 1 : 0;
        memcpy(p, rec + 32, (size_t)nl);
        p += nl;
        *p++ = '|';
        memcpy(p, contig, clen);
        p += clen;
        *p++ = '|';
        p += snprintf(p, 12, "%d", pos + 1);
        *p++ = '\n';
        const uint8_t *sq = rec + 32 + l_name + 4 * (size_t)n_cig;
        for (int i = 0; i < lseq; i++) *p++ = SEQ[(i & 1) ? (sq[i >> 1] & 0xf) : (sq[i >> 1] >> 4)];
        *p++ = '\n';
    }
    *out_len = (size_t)(p - out);
    return TDT_OK;
}


// ---- split-read fields (tiddit_signal.pyx:11-145: find_SA_query_range + SA_analysis) for selected records carrying an SA tag.
// Everything numeric is done here; the two string decisions (the order of the two contig NAMES and the swap that follows from it,
// :118-140) stay with the caller, which gets the SA contig as a byte range of `raw`.  Only the well-formed case is handled —
// entry 0 of the tag `rname,pos,strand,CIGAR,mapQ[,NM]` with unsigned decimal pos / mapQ, strand + or -, and a CIGAR of decimal
// lengths with the letters M S H D I (the only ones the reference's table knows, :23); anything else gets status 2 and the caller
// runs the literal Python, which reproduces the reference's behaviour there (KeyError on other CIGAR letters, int() oddities).

static bool sp_uint(const uint8_t *p, const uint8_t *e, long long *v) {
    if (p >= e || e - p > 9) return false;
    long long x = 0;
    for (; p < e; p++) {
        if (*p < '0' || *p > '9') return false;
        x = x * 10 + (*p - '0');
    }
    *v = x;
    return true;
}

// one selected record r (see tdt_split_fields below)
void tdt_split_one(const uint8_t *meta, const uint32_t *raw_end, const uint8_t *raw, size_t raw_len, uint32_t r, int min_q, TdtSplitOut &o) {
    memset(&o, 0, sizeof(o));
    o.status = 2;
    const size_t rec0 = r ? raw_end[r - 1] : 0u, rec1 = raw_end[r];
    int32_t pos, end, sa_rel;
    uint16_t flag;
    memcpy(&pos, meta + (size_t)r * 28 + 8, 4);
    memcpy(&end, meta + (size_t)r * 28 + 12, 4);
    memcpy(&sa_rel, meta + (size_t)r * 28 + 20, 4);
    memcpy(&flag, meta + (size_t)r * 28 + 24, 2);
    if (sa_rel < 0 || rec0 + (size_t)sa_rel >= rec1 || rec1 > raw_len) return;
    const uint8_t *s = raw + rec0 + sa_rel, *lim = raw + rec1;
    const uint8_t *e = s;
    while (e < lim && *e && *e != ';') e++;               // entry 0 of the tag (:36-39 only ever look at it)
    if (e >= lim) return;
    const uint8_t *f[6];
    int nf = 0;
    f[nf++] = s;
    for (const uint8_t *p = s; p < e && nf < 6; p++)
        if (*p == ',') f[nf++] = p + 1;
    if (nf < 5) return;
    auto fend = [&](int i) { return i + 1 < nf ? f[i + 1] - 1 : e; };
    // a sixth comma would make fend(5) wrong only for a field nobody reads (NM); fields 0..4 end at the next comma
    long long sa_pos, sa_mapq;
    if (!sp_uint(f[1], fend(1), &sa_pos) || !sp_uint(f[4], fend(4), &sa_mapq)) return;
    if (fend(2) - f[2] != 1 || (f[2][0] != '+' && f[2][0] != '-')) return;
    if (f[0] >= fend(0)) return;
    bool ascii = true;
    for (const uint8_t *p = f[0]; p < fend(0); p++) ascii = ascii && *p < 0x80;
    if (!ascii) return;
    // the SA CIGAR (:17-27): reference length (M, D), leading soft clip (hard clips skipped), well-formedness
    long long ref = 0, lead = 0;
    bool leading = true, ok = f[3] < fend(3);
    for (const uint8_t *p = f[3]; ok && p < fend(3);) {
        const uint8_t *q = p;
        while (q < fend(3) && *q >= '0' && *q <= '9') q++;
        long long len;
        if (q == p || q >= fend(3) || !sp_uint(p, q, &len)) { ok = false; break; }
        const uint8_t op = *q;
        if (op != 'M' && op != 'S' && op != 'H' && op != 'D' && op != 'I') { ok = false; break; }
        if (op == 'M' || op == 'D') ref += len;
        if (leading) {
            if (op == 'S') lead += len;
            else if (op != 'H') leading = false;
        }
        p = q + 1;
    }
    if (!ok || sa_pos + ref > 0x7fffffffll) return;
    // the read's own leading soft clip (query_alignment_start)
    const uint8_t *rec = raw + rec0 + 4;
    const int l_name = rec[8];
    uint16_t n_cig;
    memcpy(&n_cig, rec + 12, 2);
    if (rec + 32 + l_name + 4 * (size_t)n_cig > lim) return;
    long long rlead = 0;
    for (unsigned j = 0; j < n_cig; j++) {
        uint32_t cw;
        memcpy(&cw, rec + 32 + l_name + 4 * (size_t)j, 4);
        const unsigned op = cw & 0xf;
        if (op == 5) continue;
        if (op == 4) rlead += cw >> 4;
        else break;
    }
    o.chr_off = (uint32_t)(f[0] - raw);
    o.chr_len = (uint32_t)(fend(0) - f[0]);
    if (sa_mapq < min_q) {                               // :40-41
        o.status = 0;
        return;
    }
    const int seg_start = (int)sa_pos, seg_end = (int)(sa_pos + (ref ? ref : 1));
    const bool clip_before = lead < rlead, rev = (flag & 0x10) != 0, sa_minus = f[2][0] == '-';
    const int read_start = pos + 1, read_end = end + 1;
    o.read_start = read_start;
    o.read_end = read_end;
    o.split_pos = clip_before ? (rev ? read_end : read_start) : (rev ? read_start : read_end);
    o.sa_split = clip_before ? (sa_minus ? seg_start : seg_end) : (sa_minus ? seg_end : seg_start);
    o.seg_start = seg_start;
    o.seg_end = seg_end;
    o.is_reverse = rev;
    o.sa_minus = sa_minus;
    o.status = 1;
}

extern "C" int tdt_split_fields(const void *meta_, const uint32_t *raw_end, const uint8_t *raw, size_t raw_len, const uint32_t *which, size_t m,
                                int min_q, void *out_) {
    if (m && (!meta_ || !raw_end || !raw || !which || !out_)) {
        tdt_set_error("tdt_split_fields: bad argument");
        return TDT_E_ARG;
    }
    static_assert(sizeof(TdtSplitOut) == 40, "TdtSplitOut layout");
    const uint8_t *meta = (const uint8_t *)meta_;
    TdtSplitOut *out = (TdtSplitOut *)out_;
    for (size_t k = 0; k < m; k++) tdt_split_one(meta, raw_end, raw, raw_len, which[k], min_q, out[k]);
    return TDT_OK;
}
