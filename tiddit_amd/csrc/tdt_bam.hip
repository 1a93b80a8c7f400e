// Host-side BAM record decoder (no device code): turns the uncompressed BAM record stream into the
// packed SoA arrays the kernels consume.  The reference gets these fields through pysam attribute
// access one read at a time (read.reference_start / reference_end / mapq / flag / next_reference_id /
// isize ..., __main__.py:229-240, tiddit_signal.pyx:169-221); this walks the records in C.
// reference_end follows htslib's bam_endpos(): pos + sum of M/D/N/=/X lengths, pos + 1 when that sum is 0.
#include "tdt_common.h"

#include <atomic>
#include <thread>
#include <vector>


static inline uint32_t rd_u32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline int32_t rd_i32(const uint8_t *p) {
    int32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint16_t rd_u16(const uint8_t *p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}

// size in bytes of one aux value of type `t` at p (p points at the value), or -1 if malformed
static long aux_size(uint8_t t, const uint8_t *p, const uint8_t *end) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        case 'Z': case 'H': {
            const uint8_t *q = p;
            while (q < end && *q) q++;
            return q < end ? (long)(q - p) + 1 : -1;
        }
        case 'B': {
            if (p + 5 > end) return -1;
            const uint8_t st = p[0];
            const uint32_t cnt = rd_u32(p + 1);
            long es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : -1;
            return es < 0 ? -1 : 5 + es * (long)cnt;
        }
        default: return -1;
    }
}

extern "C" int tdt_bam_decode(const uint8_t *buf, size_t len, size_t max_records, size_t *consumed, size_t *n_records,
                              int32_t *tid, int32_t *pos, int32_t *end, uint8_t *mapq, uint16_t *flag, int32_t *mate_tid,
                              int32_t *mate_pos, int32_t *tlen, int32_t *l_seq, uint32_t *cigar_first, uint32_t *cigar_last,
                              uint64_t *rec_off, int64_t *sa_off) {
    if (!buf || !consumed || !n_records) {
        tdt_set_error("tdt_bam_decode: bad argument");
        return TDT_E_ARG;
    }
    // phase 1 (serial): hop the block_size chain to find the whole records in the buffer
    std::vector<size_t> offs;
    offs.reserve(len / 200 + 16);
    size_t o = 0, n = 0;
    while (n < max_records && o + 4 <= len) {
        const uint32_t bs = rd_u32(buf + o);
        if (bs < 32) {
            tdt_set_error("tdt_bam_decode: record %zu has block_size %u < 32 (corrupt stream)", n, bs);
            return TDT_E_ARG;
        }
        if (o + 4 + (size_t)bs > len) break;  // partial record: caller supplies more bytes
        offs.push_back(o);
        o += 4 + (size_t)bs;
        n++;
    }
    // phase 2 (threads): field extraction, CIGAR walk and SA lookup per record
    std::atomic<long> bad{-1};
    auto work = [&](size_t r0, size_t r1) {
        for (size_t i = r0; i < r1; i++) {
            const size_t ro = offs[i];
            const uint32_t bs = rd_u32(buf + ro);
            const uint8_t *r = buf + ro + 4;
            const int32_t p = rd_i32(r + 4);
            const uint8_t l_name = r[8];
            const uint16_t n_cig = rd_u16(r + 12);
            const int32_t lseq = rd_i32(r + 16);
            const size_t var = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)(lseq < 0 ? 0 : lseq) + 1) / 2 + (size_t)(lseq < 0 ? 0 : lseq);
            if (lseq < 0 || var > bs) {
                long exp = -1;
                bad.compare_exchange_strong(exp, (long)i);
                return;
            }
            const uint8_t *cig = r + 32 + l_name;
            int64_t rlen = 0;
            for (uint16_t k = 0; k < n_cig; k++) {
                const uint32_t c = rd_u32(cig + 4 * k);
                const uint32_t op = c & 0xf;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;  // M D N = X consume the reference
            }
            const uint16_t fl = rd_u16(r + 14);
            if ((fl & 0x4) || rlen == 0) rlen = 1;  // bam_endpos
            if (tid) tid[i] = rd_i32(r);
            if (pos) pos[i] = p;
            if (end) end[i] = (int32_t)(p + rlen);
            if (mapq) mapq[i] = r[9];
            if (flag) flag[i] = fl;
            if (mate_tid) mate_tid[i] = rd_i32(r + 20);
            if (mate_pos) mate_pos[i] = rd_i32(r + 24);
            if (tlen) tlen[i] = rd_i32(r + 28);
            if (l_seq) l_seq[i] = lseq;
            if (cigar_first) cigar_first[i] = n_cig ? rd_u32(cig) : 0xffffffffu;
            if (cigar_last) cigar_last[i] = n_cig ? rd_u32(cig + 4 * (n_cig - 1)) : 0xffffffffu;
            if (rec_off) rec_off[i] = ro;
            if (sa_off) {  // offset (from buf) of the SA:Z value, -1 when absent   (read.has_tag("SA"), tiddit_signal.pyx:199)
                int64_t found = -1;
                const uint8_t *a = r + var, *aend = r + bs;
                while (a + 3 <= aend) {
                    const uint8_t t = a[2];
                    const long sz = aux_size(t, a + 3, aend);
                    if (sz < 0 || a + 3 + sz > aend) break;
                    if (a[0] == 'S' && a[1] == 'A' && t == 'Z') {
                        found = (int64_t)((a + 3) - buf);
                        break;
                    }
                    a += 3 + sz;
                }
                sa_off[i] = found;
            }
        }
    };
    int threads = tdt_host_thread_count();
    if (n < 65536) threads = 1;
    if (threads == 1) work(0, n);
    else {
        std::vector<std::thread> pool;
        const size_t per = (n + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            const size_t r0 = (size_t)t * per, r1 = r0 + per < n ? r0 + per : n;
            if (r0 < r1) pool.emplace_back(work, r0, r1);
        }
        for (auto &t : pool) t.join();
    }
    if (bad.load() >= 0) {
        tdt_set_error("tdt_bam_decode: record %ld is inconsistent (fixed + variable fields exceed block_size)", bad.load());
        return TDT_E_ARG;
    }
    *consumed = o;
    *n_records = n;
    return TDT_OK;
}

// ---- library statistics: the sampling loop of tiddit_stats.statistics (tiddit_stats.py:17-47) over decoded field arrays (host).
// state[0] n_sampled, [1] sum of read lengths, [2] read lengths counted, [3] innie, [4] outtie, [5] done (n_sampled > n_reads).
// Insert sizes of the pairs that pass every test are appended to out_tlen (room for n), *n_out = how many.
extern "C" int tdt_stats_scan(const int32_t *tid, const int32_t *pos, const int32_t *mate_tid, const int32_t *mate_pos, const int32_t *tlen,
                              const int32_t *l_seq, const uint16_t *flag, const uint8_t *mapq, size_t n, int64_t n_reads, int min_mapq,
                              int64_t max_ins_len, int64_t *state, int32_t *out_tlen, size_t *n_out) {
    if (!state || !n_out || (n && (!tid || !pos || !mate_tid || !mate_pos || !tlen || !l_seq || !flag || !mapq || !out_tlen))) {
        tdt_set_error("tdt_stats_scan: bad argument");
        return TDT_E_ARG;
    }
    size_t k = 0;
    int64_t sampled = state[0], sum_len = state[1], n_len = state[2], innie = state[3], outtie = state[4];
    bool done = state[5] != 0;
    for (size_t i = 0; i < n && !done; i++) {
        if (tid[i] < 0) continue;                                   // samfile.fetch() skips the unplaced tail (:17)
        sum_len += l_seq[i];                                        // read_length.append(read.query_length) (:19)
        n_len++;
        sampled++;
        if (sampled > n_reads) {                                    // :22-23
            done = true;
            break;
        }
        const unsigned f = flag[i];
        if (f & 0x8u) continue;                                     // mate_is_unmapped (:25)
        if (((f & 0x10u) != 0) == ((f & 0x20u) != 0)) continue;      // is_reverse == mate_is_reverse (:28)
        if (mate_tid[i] != tid[i] || (int64_t)tlen[i] > max_ins_len) continue;   // (:31)
        if (mate_pos[i] < pos[i]) continue;                         // (:34)
        if ((f & 0xd00u) || (int)mapq[i] < min_mapq) continue;      // supplementary / secondary / duplicate / mapq (:37)
        out_tlen[k++] = tlen[i];                                    // (:40)
        if ((f & 0x10u) && !(f & 0x20u)) outtie++;                  // (:42-45)
        else innie++;
    }
    state[0] = sampled; state[1] = sum_len; state[2] = n_len; state[3] = innie; state[4] = outtie; state[5] = done ? 1 : 0;
    *n_out = k;
    return TDT_OK;
}
