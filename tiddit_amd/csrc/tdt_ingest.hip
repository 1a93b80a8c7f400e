// BAM ingest on the MI355X: compressed BGZF blocks in, packed per-read arrays out — everything between the file and
// the histogram / signal kernels stays in HBM.
//
// The reference iterates `samfile.fetch(until_eof=True)` and reads one attribute at a time (__main__.py:229-240,
// tiddit_signal.pyx:169-221); csrc/tdt_bam.hip does that walk in host C.  Here, per batch of blocks:
//   1. bgzf_inflate / bgzf_crc32 (tdt_inflate.hip) inflate the blocks behind the carried partial record;
//   2. bam_find_first + bam_find_records: the block_size chain is a serial pointer chase, so the stream is cut into 16 KiB segments;
//      a WAVE per segment tests its offsets 64 at a time for the first that passes the record sanity checks, then ONE LANE per
//      segment follows the chain from there (trying the next candidate if it breaks) until it runs cleanly to the end of the
//      segment (what BAM split guessers do), recording first / exit / count and, per record of that chain, where it starts (a
//      16-bit offset in the segment's row of `rel`);
//   3. the host walks the segment table from the known first record: the chain is accepted only if every segment's
//      guess equals the exit of its predecessor — then it is exactly the sequential decode, not a heuristic.  On any
//      disagreement the batch is copied back once and the chain is chased serially on the host (still exact);
//   4. bam_decode_fields: one WAVE per accepted segment, one lane per record (its start comes from `rel`), writes the same
//      thirteen arrays as tdt_bam_decode (bam_endpos for `end`, the SA:Z offset, first/last CIGAR op, ...); after a host
//      chase bam_decode_fields_serial walks the records of a segment with one lane instead.
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_ingest;
const char *const tdt_variant_ingest = ""
#ifdef ING_GAP
    " ING_GAP"
#endif
#ifdef ING_SEG
    " ING_SEG"
#endif
    ;

#include "tdt_common.h"
#include <mutex>

#include <algorithm>

#ifndef ING_SEG
#define ING_SEG 16384
#endif
#define ING_NONE 0xffffffffu
#define ING_EDGES 8191                       // contig runs of one batch the device reports (more: the caller derives them from the tid column)
#define ING_MAXREC ((ING_SEG + 35) / 36)      // a record is at least 4 + 32 bytes long

__device__ __forceinline__ unsigned ld_u32(const unsigned char *p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ unsigned ld_u16(const unsigned char *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }

// Sanity of the record that would start at p.  0 = complete and plausible (bs set), 1 = runs past the end of the batch,
// 2 = not a record.  DEEP adds the per-character name check and the CIGAR/l_seq identity (used on a guessed first record and
// on the record the chain lands on; records in between are pinned by the chain itself).
template <bool DEEP>
__device__ __forceinline__ int rec_check(const unsigned char *buf, long long p, long long T, int n_ref, unsigned *bs_out) {
    if (p + 4 > T) return 1;
    const unsigned bs = ld_u32(buf + p);
    if (bs < 32 || bs > (1u << 28)) return 2;
    if (p + 36 > T) return 1;
    const unsigned char *r = buf + p + 4;
    const int tid = (int)ld_u32(r), pos = (int)ld_u32(r + 4), lseq = (int)ld_u32(r + 16), mtid = (int)ld_u32(r + 20), mpos = (int)ld_u32(r + 24);
    const unsigned l_name = r[8], n_cig = ld_u16(r + 12);
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || lseq < 0 || l_name == 0) return 2;
    const unsigned long long var = 32ull + l_name + 4ull * n_cig + ((unsigned long long)lseq + 1) / 2 + (unsigned long long)lseq;
    if (var > bs) return 2;
    if (p + 36 + (long long)l_name <= T) {                      // read name: [!-~]* NUL  (SAM spec 1.4)
        if (r[32 + l_name - 1] != 0) return 2;
        for (unsigned k = 0; DEEP && k + 1 < l_name; k++) {
            const unsigned char ch = r[32 + k];
            if (ch < 33 || ch > 126) return 2;
        }
    }
    if (p + 4 + (long long)bs > T) return 1;
    if (DEEP && n_cig && lseq) {                               // query-consuming CIGAR ops spell out l_seq
        const unsigned char *cig = r + 32 + l_name;
        unsigned long long qlen = 0;
        for (unsigned k = 0; k < n_cig; k++) {
            const unsigned cw = ld_u32(cig + 4 * k), op = cw & 0xf;
            if (op > 8) return 2;
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += cw >> 4;   // M I S = X
        }
        if (qlen != (unsigned long long)lseq) return 2;
    }
    *bs_out = bs;
    return 0;
}

// Where could the first record of a segment start?  A WAVE per segment tests 64 consecutive offsets at once (one or two cache lines per
// step) for the deep record check and reports the first that passes: bam_find_records then starts its lane's search there.  Left to
// the lane itself — one candidate per dependent load, ~135 of them on average and as many as the longest record of the wave's 64
// segments for the wave — the scan was 60 % of bam_find_records (0.55 of 0.92 ms per 1.3-GB batch; timed by running the search a
// second time with its own answers as hints).  hint: ING_NONE = nothing to say (the first record's offset is known, or the segment is
// outside the search), ING_NONE - 1 = no offset of the segment passes, else the offset.
__global__ __launch_bounds__(256) void bam_find_first(const unsigned char *__restrict__ buf, long long T, long long s0, long long limit, int n_ref,
                                                      int nseg, unsigned *__restrict__ hint) {
    const int g = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (g >= nseg) return;
    const long long lo = (long long)g * ING_SEG;
    const long long hi = lo + ING_SEG < T ? lo + ING_SEG : T;
    long long p = lo;
    if (s0 >= hi || lo >= limit) p = hi;
    else if (s0 > lo) p = s0;
    const bool forced = s0 >= lo && s0 < hi;
    const long long stop = hi < limit ? hi : limit;
    unsigned h = ING_NONE;
    if (!forced && p < stop) {
        h = ING_NONE - 1;
        for (; p < stop; p += 64) {
            const long long pl = p + lane;
            unsigned bs = 0;
            const int rc = pl < stop ? rec_check<true>(buf, pl, T, n_ref, &bs) : 2;
            const unsigned long long m = __ballot(rc != 2);
            if (m) {
                h = (unsigned)(p + __builtin_ctzll(m));
                break;
            }
        }
    }
    if (lane == 0) hint[g] = h;
}

// s0 = offset of the first record when it is known (after the header / a carried record), -1 when the batch starts
// somewhere inside a file (sharded read).  limit = records starting at or beyond it belong to the next shard: a chain
// stops at the first such offset (reported as the segment's exit) and they are not counted.
__global__ __launch_bounds__(64) void bam_find_records(const unsigned char *__restrict__ buf, long long T, long long s0, long long limit,
                                                       int n_ref, int nseg, unsigned *__restrict__ first, unsigned *__restrict__ exitp,
                                                       unsigned *__restrict__ count, unsigned short *__restrict__ rel, const unsigned *__restrict__ hint) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nseg) return;
    const long long lo = (long long)g * ING_SEG;
    const long long hi = lo + ING_SEG < T ? lo + ING_SEG : T;
    unsigned f = ING_NONE, e = 0, c = 0;
    bool weak = false;                                          // a candidate that is only "a record running past the batch end"
    long long p = lo;
    if (s0 >= hi || lo >= limit) p = hi;                        // segment lies inside the header / wholly in the next shard
    else if (s0 > lo) p = s0;
    const bool forced = s0 >= lo && s0 < hi;                    // the first record's offset is known, not guessed
    const long long stop = hi < limit ? hi : limit;
    {   // bam_find_first has tested this segment's offsets 64 at a time: start at the first that passed (none: nothing to search)
        const unsigned h = hint[g];
        if (h == ING_NONE - 1) p = stop;
        else if (h != ING_NONE && (long long)h > p) p = h;
    }
    // Two phases per round so that the lanes of a wave stay in step: (A) every lane scans to its next candidate, (B) every
    // lane follows its candidate's chain.  (One fused loop made each lane's chain run while the other 63 waited.)
    bool done = p >= stop;
    while (__any(!done)) {
        unsigned bs = 0;
        int rc = 2;
        if (!done) {                                            // (A) next offset that passes the deep check (or the forced one)
            for (; p < stop; p++) {
                rc = rec_check<true>(buf, p, T, n_ref, &bs);
                if (rc != 2 || forced) break;
            }
            if (p >= stop) done = true;
        }
        if (!done) {                                            // (B) its chain to the end of the segment
            long long q = p;
            unsigned n = 0;
            unsigned short *const row = rel + (size_t)g * ING_MAXREC;   // where the records of this segment start, relative to it: the decode
            while (rc == 0 && q < stop) {                               // kernel then takes a record per LANE (a later chain overwrites a failed one)
                row[n] = (unsigned short)(q - lo);
                q += 4 + (long long)bs;
                n++;
                if (q >= T) break;                              // the batch ends exactly on a record boundary
                if (q < stop) rc = rec_check<false>(buf, q, T, n_ref, &bs);
                else rc = rec_check<true>(buf, q, T, n_ref, &bs);   // where the chain LANDS beyond the segment: must look like a record
            }
            if (rc != 2 || forced) {                            // chain ran to the segment end (or into the batch tail)
                if (n == 0 && !forced) {                        // nothing complete: remember the first such offset, keep looking
                    if (!weak) {
                        weak = true;
                        f = e = (unsigned)p;
                    }
                    p++;
                } else {
                    f = (unsigned)p;
                    e = (unsigned)q;
                    c = n;
                    if (forced && rc == 2) f = ING_NONE - 1;    // corrupt record on the true chain: never matches the walk
                    done = true;
                }
            } else p++;
            if (p >= stop) done = true;
        }
    }
    first[g] = f;
    exitp[g] = e;
    count[g] = c;
}

struct IngestOut {
    int32_t *tid, *pos, *end, *mate_tid, *mate_pos, *tlen, *l_seq;
    uint8_t *mapq;
    uint16_t *flag;
    uint32_t *cigar_first, *cigar_last;
    uint64_t *rec_off;
    int64_t *sa_off;
    unsigned long long *packed;      // cov_pack_record of (pos, end, mapq, flag) — or, when the reader is bound to a histogram
                                     // (tdt_ingest_bin_for), cov_bin_record for its bin size: what the coverage kernels read
    CovBinSpec bin;                  // bin.z != 0: binned records
};

__device__ __forceinline__ long long aux_value_size(unsigned char t, const unsigned char *p, const unsigned char *end) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        case 'Z': case 'H': {
            const unsigned char *q = p;
            while (q < end && *q) q++;
            return q < end ? (long long)(q - p) + 1 : -1;
        }
        case 'B': {
            if (p + 5 > end) return -1;
            const unsigned char st = p[0];
            const long long es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : -1;
            return es < 0 ? -1 : 5 + es * (long long)ld_u32(p + 1);
        }
        default: return -1;
    }
}

// one record at byte p of the batch -> element i of the field arrays; returns its block_size
__device__ __forceinline__ unsigned bam_decode_one(const unsigned char *__restrict__ buf, long long p, size_t i, const IngestOut &O) {
    const unsigned bs = ld_u32(buf + p);
    const unsigned char *r = buf + p + 4;
    const int pos = (int)ld_u32(r + 4), lseq = (int)ld_u32(r + 16);
    const unsigned l_name = r[8], n_cig = ld_u16(r + 12), fl = ld_u16(r + 14);
    const unsigned char *cig = r + 32 + l_name;
    long long rlen = 0;
    for (unsigned j = 0; j < n_cig; j++) {
        const unsigned cw = ld_u32(cig + 4 * j), op = cw & 0xf;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cw >> 4;   // M D N = X consume the reference
    }
    if ((fl & 0x4) || rlen == 0) rlen = 1;                                          // bam_endpos
    const int tid_ = (int)ld_u32(r);
    O.tid[i] = tid_;
    O.pos[i] = pos;
    O.end[i] = (int)(pos + rlen);
    O.mapq[i] = r[9];
    O.flag[i] = (uint16_t)fl;
    if (O.bin.z) {
        const int nb_ = (tid_ >= 0 && tid_ < O.bin.n_contigs) ? O.bin.d_nbins[tid_] : 0;          // unplaced reads: no bins (never pushed)
        O.packed[i] = cov_bin_record(pos, (int)(pos + rlen), r[9], fl, nb_, O.bin.z, O.bin.magic, O.bin.shift, O.bin.mode1 != 0);
    } else {
        O.packed[i] = cov_pack_record(pos, (int)(pos + rlen), r[9], fl);
    }
    O.mate_tid[i] = (int)ld_u32(r + 20);
    O.mate_pos[i] = (int)ld_u32(r + 24);
    O.tlen[i] = (int)ld_u32(r + 28);
    O.l_seq[i] = lseq;
    O.cigar_first[i] = n_cig ? ld_u32(cig) : 0xffffffffu;
    O.cigar_last[i] = n_cig ? ld_u32(cig + 4 * (n_cig - 1)) : 0xffffffffu;
    O.rec_off[i] = (uint64_t)p;
    long long found = -1;                                                           // SA:Z value offset (tiddit_signal.pyx:199)
    const unsigned char *a = r + 32 + l_name + 4 * n_cig + ((size_t)lseq + 1) / 2 + (size_t)lseq, *aend = r + bs;
    while (a + 3 <= aend) {
        const unsigned char t = a[2];
        const long long sz = aux_value_size(t, a + 3, aend);
        if (sz < 0 || a + 3 + sz > aend) break;
        if (a[0] == 'S' && a[1] == 'A' && t == 'Z') {
            found = (long long)((a + 3) - buf);
            break;
        }
        a += 3 + sz;
    }
    O.sa_off[i] = found;
    return bs;
}

// A WAVE per segment, a LANE per record: bam_find_records left the start of every record of its segment's chain in `rel`, so the
// records of a segment are decoded side by side and their fields leave in coalesced stores.  (One lane per segment walking its ~55
// records — rounds 2-4 — was 1 250 waves of 64 dependent chains for a 1.3-GB batch: a sixth of the chip's wave slots, every load a
// round trip nobody hid.)
__global__ __launch_bounds__(256) void bam_decode_fields(const unsigned char *__restrict__ buf, long long T, int nseg,
                                                         const unsigned *__restrict__ base, const unsigned *__restrict__ count,
                                                         const unsigned short *__restrict__ rel, IngestOut O) {
    const int g = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (g >= nseg) return;
    const unsigned n = count[g], b = base[g];
    if (b == ING_NONE || n == 0) return;
    const long long lo = (long long)g * ING_SEG;
    const unsigned short *const row = rel + (size_t)g * ING_MAXREC;
    for (unsigned k = (unsigned)lane; k < n; k += 64) (void)bam_decode_one(buf, lo + row[k], (size_t)b + k, O);
}

// The same with a lane per segment walking its records: after a host chase (the segment table rebuilt on the host, no `rel` rows)
__global__ __launch_bounds__(64) void bam_decode_fields_serial(const unsigned char *__restrict__ buf, long long T, int nseg,
                                                               const unsigned *__restrict__ first, const unsigned *__restrict__ base,
                                                               const unsigned *__restrict__ count, IngestOut O) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nseg) return;
    const unsigned n = count[g];
    if (base[g] == ING_NONE || n == 0) return;
    long long p = first[g];
    size_t i = base[g];
    for (unsigned k = 0; k < n; k++, i++) p += 4 + (long long)bam_decode_one(buf, p, i, O);
}

// positions where tid changes (i = 0 included): the per-contig runs of a coordinate-sorted batch
// (the contig id at every edge rides along: a human reference with its alt / decoy / HLA contigs has thousands of runs per batch, and a
// 4-byte copy per run to learn its tid was 40 ms of a 25-Mb job)
__global__ void bam_tid_edges(const int32_t *__restrict__ tid, size_t n, unsigned *__restrict__ edges, unsigned cap, unsigned *__restrict__ n_edges) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t t = tid[i];
    if (i == 0 || t != tid[i - 1]) {
        const unsigned k = atomicAdd(n_edges, 1u);
        if (k < cap) {
            edges[k] = (unsigned)i;
            edges[cap + 1 + k] = (unsigned)t;
        }
    }
}

#ifndef ING_GAP
#define ING_GAP (1u << 20)
#endif

struct tdt_ingest {
    tdt_ctx *ctx = nullptr;
    int n_ref = 0;
    tdt_buf seg, soa;                              // device buffers (grow only)
    // A span's inflate no longer depends on the batch before it.  Span s lives in slot[s % NSLOT]: its compressed blocks, its block table
    // and its OUTPUT, which the inflate kernel writes `gap` bytes into the buffer — the partial record the previous batch ended with (known
    // only when that batch has been parsed) is copied in front of it afterwards, by the second half of the push.  Three slots: while batch k
    // is being parsed and consumed on the launch stream, span k+1 inflates and the consumers of batch k-1 may still be running.
    enum { NSLOT = 3 };
    struct Slot {
        tdt_buf out, comp, table;
        void *h_table = nullptr;                   // pinned staging of the block table when the reader thread did not bring it along
        size_t h_table_cap = 0;
        unsigned *h_summary = nullptr;             // pinned: {first failed block, failed blocks} of the span
        hipEvent_t inflated = nullptr;             // inflate stream: the span is inflated and checked, its status word is on the host
        hipEvent_t released = nullptr;             // launch stream: everything that reads the batch that lived here was enqueued before it
        bool released_set = false;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;   // inflate stream: around the span's copy (when the push makes it), behind inflate + CRC
        hipEvent_t hit_t0 = nullptr, hit_t1 = nullptr;         // the prefetch slot's events around the copy it made
        const uint8_t *host = nullptr;             // the span begun here and not pushed yet (null: none)
        size_t len = 0, produced = 0, nb = 0;
        unsigned *d_status = nullptr;
        double table_ms = 0;
        bool prefetched = false;
    } slot[NSLOT];
    unsigned long long seq_begun = 0, seq_pushed = 0;   // spans whose first half was enqueued / whose push has run
    int cur = -1;                                  // slot of the current batch (-1: none yet)
    size_t lead = 0;                               // the current batch starts at slot[cur].out + lead  (= gap - carried bytes)
    size_t gap = ING_GAP;
    tdt_buf carrybuf;                              // the partial record behind the current batch
    hipStream_t inf_stream[2] = {nullptr, nullptr};     // spans alternate: the head of span k+1 fills the wave slots the tail of span k leaves
    int reserve = 0;                               // workgroups per CU the inflate grid leaves free (TIDDIT_INFLATE_RESERVE).  0: measured best —
                                                   // the launch stream's kernels get the wave slots the previous span's tail frees, and a
                                                   // grid one workgroup per CU short costs the inflate more than they gain (profiles/r06_sv_reserve_3000mb.txt)
    // compressed blocks of spans copied ahead by tdt_ingest_prefetch: the copy of span k+1 can be issued while span k has not been begun
    struct Prefetch {
        tdt_buf buf;
        const uint8_t *host = nullptr;             // what the slot holds: host pointer / length of the span (null = free)
        size_t len = 0;
        hipEvent_t done = nullptr;
        hipEvent_t t0 = nullptr, t1 = nullptr;     // around the copy, for tdt_ingest_timing
        std::vector<BzDesc> blocks;                // the span's BGZF block table, built by the thread that read it ...
        size_t produced = 0;
        tdt_buf table;                             // ... and already on the device (same layout as Slot::table)
    } pf[3];                                       // spans the reader can be ahead by: queued, held back by the full queue, taken but not begun yet
    std::mutex pf_mu;                              // the slots are filled by the reader thread (tdt_ingest_prefetch) and emptied by the pushing one
    tdt_buf pin;                                   // pinned staging for the segment table / edges
    size_t carry = 0;                              // bytes of the partial record in carrybuf
    size_t out_len = 0;                            // carry + inflated bytes of the current batch
    size_t n_records = 0, rec_cap = 0;
    IngestOut O{};
    std::vector<unsigned> edges;
    std::vector<int32_t> edge_tids;              // the contig id of the run that starts at edges[k]
    bool edges_overflow = false;
    bool failed = false;                           // a push returned an error: the stream position is undefined from then on
    size_t host_chases = 0;                        // batches whose record chain had to be chased on the host
    // where the last push spent its time (tdt_ingest_timing): the slot's events on the inflate stream, these on the launch stream, host clocks
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};   // 0-1 record search, 2-3 field decode
    double t_chain_ms = 0, t_wall_ms = 0;
    bool t_have_decode = false, t_have_find = false;
};

static double ing_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// Device buffers of the ingest come from — and go back to — a process-wide cache instead of hipMalloc / hipFree: a reader's buffers are
// hundreds of megabytes to gigabytes, `tiddit --sv` keeps seven batches of the statistics pass (tdt_ingest_retain: a fresh 1.7-GB
// output buffer and a fresh 0.4-GB array block each) and frees them a moment later, and every pass over a file opens a reader.  GB-sized
// hipMalloc / hipFree calls take 10-40 ms apiece when the driver has to map or unmap the range, and they came in bursts: the
// statistics stage of a 240-Mb job measured 0.13 s or 0.5-0.7 s from one run to the next.
// Bounded PER DEVICE: a quarter of the device's memory, at most 64 GB (TIDDIT_INGEST_CACHE_MB overrides, 0 switches the cache off);
// a request takes the smallest cached buffer that is large enough and at most twice the size.  The cache is never what makes an
// allocation fail: every device allocation of the library goes through tdt_dev_malloc, which returns the cache to the driver
// (tdt_dev_cache_flush, also behind the ABI as tdt_device_cache_flush) and tries again; the last context of a device to be destroyed
// flushes it too.
namespace {
struct IngCached {
    void *p;
    size_t cap;
    int device;
};
enum { ING_MAX_DEV = 64 };
std::mutex ing_cache_mu;
std::vector<IngCached> ing_cache;
size_t ing_cache_bytes[ING_MAX_DEV] = {};
size_t ing_cache_limit[ING_MAX_DEV] = {};
bool ing_cache_limit_known[ING_MAX_DEV] = {};

// (lock held; the caller has made `device` current)
size_t ing_limit(int device) {
    if (device < 0 || device >= ING_MAX_DEV) return 0;
    if (!ing_cache_limit_known[device]) {
        size_t lim = (size_t)64 << 30;
        const char *e = getenv("TIDDIT_INGEST_CACHE_MB");
        if (e && *e) {
            lim = (size_t)strtoull(e, nullptr, 10) << 20;
        } else {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                if (total_b / 4 < lim) lim = total_b / 4;
            } else {
                (void)hipGetLastError();
                lim = 0;
            }
        }
        ing_cache_limit[device] = lim;
        ing_cache_limit_known[device] = true;
    }
    return ing_cache_limit[device];
}

void *ing_dev_alloc(int device, size_t cap, size_t *got_cap) {
    {
        std::lock_guard<std::mutex> lock(ing_cache_mu);
        int best = -1;
        for (int i = 0; i < (int)ing_cache.size(); i++)
            if (ing_cache[(size_t)i].device == device && ing_cache[(size_t)i].cap >= cap && ing_cache[(size_t)i].cap <= 2 * cap + (1u << 20) &&
                (best < 0 || ing_cache[(size_t)i].cap < ing_cache[(size_t)best].cap))
                best = i;
        if (best >= 0) {
            const IngCached c = ing_cache[(size_t)best];
            ing_cache.erase(ing_cache.begin() + best);
            ing_cache_bytes[device] -= c.cap;
            *got_cap = c.cap;
            return c.p;
        }
    }
    void *p = nullptr;
    if (tdt_dev_malloc(&p, cap) != hipSuccess) return nullptr;       // (has already tried again behind a flush)
    *got_cap = cap;
    return p;
}

void ing_dev_free(int device, void *p, size_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(ing_cache_mu);
        if (device >= 0 && device < ING_MAX_DEV && cap >= (1u << 20) && ing_cache_bytes[device] + cap <= ing_limit(device)) {
            ing_cache.push_back(IngCached{p, cap, device});
            ing_cache_bytes[device] += cap;
            return;
        }
    }
    (void)hipFree(p);
}
}  // namespace

size_t tdt_dev_cache_flush(int device) {
    std::vector<IngCached> drop;
    {
        std::lock_guard<std::mutex> lock(ing_cache_mu);
        std::vector<IngCached> keep;
        for (auto &c : ing_cache) (device < 0 || c.device == device ? drop : keep).push_back(c);
        ing_cache.swap(keep);
        for (auto &c : drop) ing_cache_bytes[c.device] -= c.cap;
    }
    size_t released = 0;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto &c : drop) {
        (void)hipSetDevice(c.device);
        (void)hipFree(c.p);
        released += c.cap;
    }
    if (!drop.empty() && cur >= 0) (void)hipSetDevice(cur);
    return released;
}

size_t tdt_dev_cache_held(int device) {
    std::lock_guard<std::mutex> lock(ing_cache_mu);
    size_t n = 0;
    for (int d = 0; d < ING_MAX_DEV; d++)
        if (device < 0 || d == device) n += ing_cache_bytes[d];
    return n;
}

// Return the cached device buffers of the context's device to the driver (*released = their bytes, may be NULL).  The library does
// this by itself when one of its allocations fails; an application that shares the device with other allocators calls it between jobs.
extern "C" int tdt_device_cache_flush(tdt_ctx *ctx, uint64_t *released) {
    if (!ctx) {
        tdt_set_error("tdt_device_cache_flush: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    TDT_HIP(hipStreamSynchronize(ctx->copy_stream));
    const size_t n = tdt_dev_cache_flush(ctx->device);
    if (released) *released = (uint64_t)n;
    return TDT_OK;
}
extern "C" uint64_t tdt_device_cache_bytes(tdt_ctx *ctx) { return ctx ? (uint64_t)tdt_dev_cache_held(ctx->device) : 0; }

static int ing_grow(tdt_ingest *g, tdt_buf &b, size_t bytes, bool keep = false) {
    if (b.cap >= bytes) return TDT_OK;
    size_t cap = bytes + bytes / 4 + 4096;
    void *p = ing_dev_alloc(g->ctx->device, cap, &cap);
    if (!p) {
        tdt_set_error("tdt_ingest: device allocation of %zu bytes failed", cap);
        return TDT_E_NOMEM;
    }
    if (keep && b.p) {
        if (hipMemcpyAsync(p, b.p, b.cap, hipMemcpyDeviceToDevice, g->ctx->stream) != hipSuccess || hipStreamSynchronize(g->ctx->stream) != hipSuccess) {
            ing_dev_free(g->ctx->device, p, cap);
            tdt_set_error("tdt_ingest: device copy failed");
            return TDT_E_HIP;
        }
    } else if (b.p) {
        // the old buffer may still be read by kernels in flight on the launch stream: it leaves this reader only behind them
        if (hipStreamSynchronize(g->ctx->stream) != hipSuccess) (void)hipGetLastError();
    }
    ing_dev_free(g->ctx->device, b.p, b.cap);
    b.p = p;
    b.cap = cap;
    return TDT_OK;
}

extern "C" int tdt_ingest_create(tdt_ctx *ctx, int n_ref, tdt_ingest **out) {
    if (!ctx || !out || n_ref < 0) {
        tdt_set_error("tdt_ingest_create: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    tdt_ingest *g = new tdt_ingest();
    g->ctx = ctx;
    g->n_ref = n_ref;
    for (auto &e : g->tev)
        if (hipEventCreate(&e) != hipSuccess) {
            (void)hipGetLastError();
            e = nullptr;
        }
    if (const char *e = getenv("TIDDIT_INGEST_GAP")) {               // (tests: a gap smaller than a record exercises the relocation path)
        const long long v = atoll(e);
        if (v >= 0) g->gap = ((size_t)v + 255) & ~(size_t)255;
    }
    if (const char *e = getenv("TIDDIT_INFLATE_RESERVE")) g->reserve = atoi(e) < 0 ? 0 : atoi(e) > 7 ? 7 : atoi(e);
    unsigned *h_sum = nullptr;
    bool ok = hipHostMalloc((void **)&h_sum, 64 * tdt_ingest::NSLOT, hipHostMallocDefault) == hipSuccess;
    // the inflate streams: lowest priority, so that the kernels of the launch stream (record search, field decode, the caller's coverage
    // and signal kernels) get the wave slots that free up while a span inflates
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) (void)hipGetLastError(), prio_lo = 0;
    for (auto &st : g->inf_stream) ok = ok && hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo) == hipSuccess;
    for (int i = 0; i < tdt_ingest::NSLOT && ok; i++) {
        auto &S = g->slot[i];
        S.h_summary = h_sum + 16 * i;
        ok = ok && hipEventCreateWithFlags(&S.inflated, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&S.released, hipEventDisableTiming) == hipSuccess;
        for (hipEvent_t *e : {&S.e0, &S.e1, &S.e2})
            if (hipEventCreate(e) != hipSuccess) (void)hipGetLastError(), *e = nullptr;
    }
    if (!ok) {
        (void)hipGetLastError();
        if (h_sum && !g->slot[0].h_summary) (void)hipHostFree(h_sum);
        (void)tdt_ingest_destroy(g);
        tdt_set_error("tdt_ingest_create: streams / events / pinned memory could not be created");
        return TDT_E_NOMEM;
    }
    *out = g;
    return TDT_OK;
}

extern "C" int tdt_ingest_destroy(tdt_ingest *g) {
    if (!g) return TDT_OK;
    (void)hipSetDevice(g->ctx->device);
    for (auto &st : g->inf_stream)
        if (st) (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(g->ctx->stream);
    (void)hipStreamSynchronize(g->ctx->copy_stream);
    for (auto &p : g->pf) {
        if (p.done) (void)hipEventDestroy(p.done);
        if (p.t0) (void)hipEventDestroy(p.t0);
        if (p.t1) (void)hipEventDestroy(p.t1);
    }
    for (auto &e : g->tev)
        if (e) (void)hipEventDestroy(e);
    for (auto &S : g->slot) {
        for (hipEvent_t e : {S.inflated, S.released, S.e0, S.e1, S.e2, S.hit_t0, S.hit_t1})
            if (e) (void)hipEventDestroy(e);
        for (tdt_buf *b : {&S.out, &S.comp, &S.table}) ing_dev_free(g->ctx->device, b->p, b->cap);     // (every stream is idle: synchronised above)
        if (S.h_table) (void)hipHostFree(S.h_table);
    }
    for (tdt_buf *b : {&g->pf[0].buf, &g->pf[1].buf, &g->pf[2].buf, &g->pf[0].table, &g->pf[1].table, &g->pf[2].table, &g->seg, &g->soa, &g->carrybuf})
        ing_dev_free(g->ctx->device, b->p, b->cap);
    for (auto &st : g->inf_stream)
        if (st) (void)hipStreamDestroy(st);
    if (g->pin.p) (void)hipHostFree(g->pin.p);
    if (g->slot[0].h_summary) (void)hipHostFree(g->slot[0].h_summary);
    delete g;
    return TDT_OK;
}

// Start copying the NEXT span of blocks to the device on the context's copy stream; a following push of exactly this
// (pointer, length) uses it instead of copying, so the transfer overlaps the kernels of the push in between.
extern "C" int tdt_ingest_prefetch(tdt_ingest *g, const uint8_t *comp, size_t len) {
    if (!g || !comp || !len) {
        tdt_set_error("tdt_ingest_prefetch: bad argument");
        return TDT_E_ARG;
    }
    tdt_ctx *ctx = g->ctx;
    TDT_HIP(hipSetDevice(ctx->device));
    const size_t comp_pad = (len + 4096 + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lock(g->pf_mu);        // (may be called from the thread that reads the file while another one pushes)
    tdt_ingest::Prefetch *slot = nullptr;
    for (auto &p : g->pf)
        if (!p.host) {
            slot = &p;
            break;
        }
    if (!slot) return TDT_OK;                            // every slot holds a span that has not been pushed yet: this one is copied by its push
    // (a buffer swapped into a slot by an earlier push is free: that push waited for its inflate kernel before returning)
    if (slot->buf.cap < comp_pad) {                     // grow without touching the launch stream (ing_grow's copy path is not needed here)
        size_t cap = comp_pad + comp_pad / 4 + 4096;
        void *np_ = ing_dev_alloc(ctx->device, cap, &cap);
        if (!np_) {
            tdt_set_error("tdt_ingest_prefetch: device allocation of %zu bytes failed", cap);
            return TDT_E_NOMEM;
        }
        if (slot->buf.p) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            ing_dev_free(ctx->device, slot->buf.p, slot->buf.cap);
        }
        slot->buf.p = np_;
        slot->buf.cap = cap;
    }
    if (!slot->done) {
        TDT_HIP(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
        (void)hipEventCreate(&slot->t0);
        (void)hipEventCreate(&slot->t1);
    }
    if (slot->t0) (void)hipEventRecord(slot->t0, ctx->copy_stream);
    TDT_HIP(hipMemcpyAsync(slot->buf.p, comp, len, hipMemcpyHostToDevice, ctx->copy_stream));
    // (the closing event sits BEHIND THE COPY, in front of the pad's memset: the memset is a kernel, and while the persistent waves of the
    // previous span's inflate kernel hold every slot of the chip it waits for that kernel to end — rounds 3-4 recorded the event behind it
    // and reported "h2d" sums of 1.65-1.73 s for a 3-Gb job whose 54 GB cross PCIe in 1.0 s; tools/time_h2d_contention.py)
    if (slot->t1) (void)hipEventRecord(slot->t1, ctx->copy_stream);
    TDT_HIP(hipMemsetAsync((char *)slot->buf.p + len, 0, comp_pad - len, ctx->copy_stream));
    // the block table (a serial hop over the span's block headers: 3.6 ms per 260 MB) is built here, off the pushing thread's path, and
    // follows the span onto the device
    slot->blocks.clear();
    slot->produced = 0;
    if (tdt_bz_block_table(comp, len, slot->blocks, &slot->produced) == TDT_OK && !slot->blocks.empty()) {
        const size_t nb = slot->blocks.size();
        const size_t tab = (nb * sizeof(BzDesc) + 255) & ~(size_t)255, stb = (nb * 4 + 255) & ~(size_t)255;
        if (slot->table.cap < tab + stb + 256) {
            size_t cap = (tab + stb + 256) * 5 / 4 + 4096;
            void *np_ = ing_dev_alloc(ctx->device, cap, &cap);
            if (np_) {
                if (slot->table.p) {
                    (void)hipStreamSynchronize(ctx->copy_stream);
                    ing_dev_free(ctx->device, slot->table.p, slot->table.cap);
                }
                slot->table.p = np_;
                slot->table.cap = cap;
            } else {
                (void)hipGetLastError();
                slot->blocks.clear();
            }
        }
        if (!slot->blocks.empty() &&
            hipMemcpyAsync(slot->table.p, slot->blocks.data(), nb * sizeof(BzDesc), hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess) {
            (void)hipGetLastError();
            slot->blocks.clear();
        }
    } else {
        slot->blocks.clear();                              // (a malformed span: the push reports it)
    }
    TDT_HIP(hipEventRecord(slot->done, ctx->copy_stream));
    slot->host = comp;
    slot->len = len;
    return TDT_OK;
}

extern "C" int tdt_ingest_push(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t *n_records) {
    return tdt_ingest_push_bounded(g, comp, len, skip, (size_t)-1, n_records, nullptr, nullptr);
}

static int ing_push(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t own_bytes, size_t *n_records, size_t *first_off,
                    size_t *next_off);

extern "C" int tdt_ingest_push_bounded(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t own_bytes, size_t *n_records,
                                       size_t *first_off, size_t *next_off) {
    if (!g || (!comp && len) || !n_records) {
        tdt_set_error("tdt_ingest_push: bad argument");
        return TDT_E_ARG;
    }
    if (g->failed) {
        tdt_set_error("tdt_ingest_push: an earlier push on this stream failed; create a new tdt_ingest");
        return TDT_E_ARG;
    }
    const double t0 = ing_now_ms();
    const int rc = ing_push(g, comp, len, skip, own_bytes, n_records, first_off, next_off);
    g->t_wall_ms = ing_now_ms() - t0;
    if (rc != TDT_OK) g->failed = true;
    return rc;
}

// First half of a push: the span's bytes and block table on the device (taken from the prefetch slot that holds them, or copied here), the
// inflate + CRC kernels and the copy of their status word — all ENQUEUED on one of the reader's inflate streams, nothing waited for, and
// nothing of it depends on the batch before: the output goes `gap` bytes into the slot's own buffer (ing_push puts the carried bytes in front).
static int ing_push_begin(tdt_ingest *g, const uint8_t *comp, size_t len) {
    tdt_ctx *ctx = g->ctx;
    TDT_HIP(hipSetDevice(ctx->device));
    const unsigned long long seq = g->seq_begun;
    tdt_ingest::Slot &S = g->slot[seq % tdt_ingest::NSLOT];
    hipStream_t sb = g->inf_stream[seq & 1];
    std::vector<BzDesc> blocks;
    size_t produced = 0;
    const double t_begin = ing_now_ms();
    // a span the reader thread prefetched brings its block table along (host copy + device copy)
    tdt_ingest::Prefetch *hit = nullptr;
    bool table_on_device = false;
    int rc = TDT_OK;
    {
        std::lock_guard<std::mutex> lock(g->pf_mu);
        for (auto &p : g->pf)
            if (p.host == comp && p.len == len && len) hit = &p;
        if (hit && !hit->blocks.empty()) {
            blocks.swap(hit->blocks);
            produced = hit->produced;
            table_on_device = true;
        }
    }
    if (!table_on_device) {
        rc = tdt_bz_block_table(comp, len, blocks, &produced);
        if (rc) return rc;
    }
    S.table_ms = ing_now_ms() - t_begin;
    S.prefetched = false;
    if (g->gap + produced >= 0xfffffff0ull - (64u << 20)) {
        tdt_set_error("tdt_ingest_push: batch inflates to %zu bytes; feed at most 3 GiB of records per call", produced);
        return TDT_E_RANGE;
    }
    // the batch that lived in this slot three spans ago: whatever reads it was enqueued on the launch stream before `released`
    if (S.released_set) TDT_HIP(hipStreamWaitEvent(sb, S.released, 0));
    S.released_set = false;
    rc = ing_grow(g, S.out, g->gap + produced + 256);
    if (rc) return rc;
    const size_t nb = blocks.size();
    unsigned *d_status = nullptr;
    S.h_summary[0] = S.h_summary[1] = 0;
    if (S.e0) (void)hipEventRecord(S.e0, sb);
    if (nb) {
        const size_t comp_pad = (len + 4096 + 255) & ~(size_t)255;
        rc = ing_grow(g, S.comp, comp_pad);
        if (rc) return rc;
        const size_t tab = (nb * sizeof(BzDesc) + 255) & ~(size_t)255, stb = (nb * 4 + 255) & ~(size_t)255;
        rc = ing_grow(g, S.table, tab + stb + 256);
        if (rc) return rc;
        BzDesc *d_blocks = (BzDesc *)S.table.p;
        d_status = (unsigned *)((char *)S.table.p + tab);
        unsigned *d_summary = (unsigned *)((char *)d_status + stb);
        if (table_on_device && hit && hit->table.cap < tab + stb + 256) table_on_device = false;       // (cannot happen: sized alike)
        bool was_hit = false;
        {
            std::lock_guard<std::mutex> lock(g->pf_mu);
            if (hit && (hit->host != comp || hit->len != len || hit->buf.cap < comp_pad)) hit = nullptr;
            if (hit) {
                // (what goes back into the prefetch slot is free: the span that used these buffers, three spans ago, has been pushed)
                std::swap(S.comp, hit->buf);                      // the span is already on the device (copy stream)
                if (table_on_device) {
                    std::swap(S.table, hit->table);               // ... and so is its block table
                    d_blocks = (BzDesc *)S.table.p;
                    d_status = (unsigned *)((char *)S.table.p + tab);
                    d_summary = (unsigned *)((char *)d_status + stb);
                }
                TDT_HIP(hipStreamWaitEvent(sb, hit->done, 0));
                std::swap(S.hit_t0, hit->t0);                     // (the slot gets the previous pair back: events are reused)
                std::swap(S.hit_t1, hit->t1);
                if (!hit->t0) (void)hipEventCreate(&hit->t0);
                if (!hit->t1) (void)hipEventCreate(&hit->t1);
                hit->host = nullptr;
                S.prefetched = true;
                was_hit = true;
            }
        }
        if (!was_hit) {
            TDT_HIP(hipMemcpyAsync(S.comp.p, comp, len, hipMemcpyHostToDevice, sb));
            TDT_HIP(hipMemsetAsync((char *)S.comp.p + len, 0, comp_pad - len, sb));
        }
        if (S.e1) (void)hipEventRecord(S.e1, sb);
        unsigned char *d_comp = (unsigned char *)S.comp.p;
        if (!(was_hit && table_on_device)) {
            // (the table's host copy must outlive the asynchronous transfer: staged through the SLOT's pinned block — a context-wide one
            // could be overwritten by another reader of the context before the copy has run)
            if (S.h_table_cap < nb * sizeof(BzDesc)) {
                if (S.h_table) (void)hipHostFree(S.h_table);
                S.h_table = nullptr;
                S.h_table_cap = 0;
                const size_t cap = nb * sizeof(BzDesc) * 5 / 4 + 4096;
                TDT_HIP(hipHostMalloc(&S.h_table, cap, hipHostMallocDefault));
                S.h_table_cap = cap;
            }
            memcpy(S.h_table, blocks.data(), nb * sizeof(BzDesc));
            TDT_HIP(hipMemcpyAsync(d_blocks, S.h_table, nb * sizeof(BzDesc), hipMemcpyHostToDevice, sb));
        }
        rc = tdt_bz_launch_on(ctx, sb, g->reserve, d_comp, d_blocks, nb, (unsigned char *)S.out.p + g->gap, true, d_status, d_summary);
        if (rc) return rc;
        if (S.e2) (void)hipEventRecord(S.e2, sb);
        TDT_HIP(hipMemcpyAsync(S.h_summary, d_summary, 8, hipMemcpyDeviceToHost, sb));
    } else {
        if (S.e1) (void)hipEventRecord(S.e1, sb);
        if (S.e2) (void)hipEventRecord(S.e2, sb);
    }
    TDT_HIP(hipEventRecord(S.inflated, sb));
    S.host = comp;
    S.len = len;
    S.produced = produced;
    S.nb = nb;
    S.d_status = d_status;
    g->seq_begun = seq + 1;
    return TDT_OK;
}

// Enqueue the first half of a coming span's push on the reader's inflate streams.  It depends on nothing the launch stream holds — the span
// inflates into its own output buffer — so it may be called as soon as the span is in host memory: before the push of the span in front of it,
// and while the current batch's consumers run.  At most two spans can be begun beyond the current batch; their pushes must come in the same order.
extern "C" int tdt_ingest_push_ahead(tdt_ingest *g, const uint8_t *comp, size_t len) {
    if (!g || !comp || !len) {
        tdt_set_error("tdt_ingest_push_ahead: bad argument");
        return TDT_E_ARG;
    }
    if (g->failed || g->seq_begun - g->seq_pushed >= 2) {
        tdt_set_error(g->failed ? "tdt_ingest_push_ahead: an earlier push on this stream failed" : "tdt_ingest_push_ahead: two spans are already inflating ahead");
        return TDT_E_ARG;
    }
    const int rc = ing_push_begin(g, comp, len);
    if (rc != TDT_OK) g->failed = true;
    return rc;
}

static int ing_push(tdt_ingest *g, const uint8_t *comp, size_t len, size_t skip, size_t own_bytes, size_t *n_records, size_t *first_off,
                    size_t *next_off) {
    const bool unknown_start = skip == (size_t)-1;
    if (unknown_start && (g->carry || g->seq_pushed)) {
        tdt_set_error("tdt_ingest_push_bounded: an unknown start is only possible on a fresh stream");
        return TDT_E_ARG;
    }
    if (unknown_start) skip = 0;
    if (first_off) *first_off = (size_t)-1;
    if (next_off) *next_off = (size_t)-1;
    tdt_ctx *ctx = g->ctx;
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = TDT_OK;
    if (g->seq_begun > g->seq_pushed) {
        const tdt_ingest::Slot &B = g->slot[g->seq_pushed % tdt_ingest::NSLOT];
        if (B.host != comp || B.len != len) {
            tdt_set_error("tdt_ingest_push: another span was started with tdt_ingest_push_ahead");
            return TDT_E_ARG;
        }
    } else {
        rc = ing_push_begin(g, comp, len);
        if (rc) return rc;
    }
    const int si = (int)(g->seq_pushed % tdt_ingest::NSLOT);
    tdt_ingest::Slot &S = g->slot[si];
    // the batch in front of this one: everything that reads it has been enqueued (its slot is inflated into again three spans on)
    if (g->cur >= 0) {
        TDT_HIP(hipEventRecord(g->slot[g->cur].released, st));
        g->slot[g->cur].released_set = true;
    }
    g->seq_pushed++;
    g->cur = si;
    S.host = nullptr;
    const size_t produced = S.produced, carry = g->carry, T = carry + produced, nb = S.nb;
    TDT_HIP(hipEventSynchronize(S.inflated));
    if (nb && S.h_summary[1]) {
        unsigned code = 0;
        TDT_HIP(hipMemcpy(&code, S.d_status + S.h_summary[0], 4, hipMemcpyDeviceToHost));
        tdt_set_error("tdt_ingest_push: %u of %zu BGZF blocks failed; first is block %u: %s", S.h_summary[1], nb, S.h_summary[0], tdt_bz_err_name(code));
        return TDT_E_ARG;
    }
    TDT_HIP(hipStreamWaitEvent(st, S.inflated, 0));
    if (T >= 0xfffffff0ull) {
        tdt_set_error("tdt_ingest_push: batch inflates to %zu bytes; feed at most 3 GiB of records per call", T);
        return TDT_E_RANGE;
    }
    // ---- the carried partial record goes in front of the span's output
    if (carry <= g->gap) {
        g->lead = g->gap - carry;
        if (carry) TDT_HIP(hipMemcpyAsync((char *)S.out.p + g->lead, g->carrybuf.p, carry, hipMemcpyDeviceToDevice, st));
    } else {
        // a record longer than the gap: the span's output moves behind it in a buffer of its own (one extra pass over the batch; the
        // default gap is 1 MB, a record that long is a very long read)
        tdt_buf moved;
        moved.p = ing_dev_alloc(ctx->device, T + 256, &moved.cap);
        if (!moved.p) {
            tdt_set_error("tdt_ingest_push: device allocation of %zu bytes failed", T + 256);
            return TDT_E_NOMEM;
        }
        hipError_t e = hipMemcpyAsync(moved.p, g->carrybuf.p, carry, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && produced) e = hipMemcpyAsync((char *)moved.p + carry, (char *)S.out.p + g->gap, produced, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            ing_dev_free(ctx->device, moved.p, moved.cap);
            tdt_set_error("tdt_ingest_push: device copy failed: %s", hipGetErrorString(e));
            return TDT_E_HIP;
        }
        ing_dev_free(ctx->device, S.out.p, S.out.cap);
        S.out = moved;
        g->lead = 0;
    }
    unsigned char *d_out = (unsigned char *)S.out.p + g->lead;
    g->out_len = T;
    g->n_records = 0;
    g->edges.clear();
    g->edge_tids.clear();
    g->edges_overflow = false;
    g->t_have_find = g->t_have_decode = false;
    g->t_chain_ms = 0;
    *n_records = 0;
    if (skip > T) {
        tdt_set_error("tdt_ingest_push: skip (%zu) exceeds the inflated bytes (%zu)", skip, T);
        return TDT_E_ARG;
    }
    if (T == skip) {
        g->carry = 0;
        return TDT_OK;
    }
    const bool bounded = own_bytes != (size_t)-1;
    const size_t limit = bounded ? carry + own_bytes : T;         // records starting at or after it are the next shard's
    if (limit > T) {
        tdt_set_error("tdt_ingest_push_bounded: own_bytes (%zu) exceeds the inflated bytes (%zu)", own_bytes, produced);
        return TDT_E_ARG;
    }
    // ---- find the records: per-segment guesses on the device, chain check on the host
    const int nseg = (int)((T + ING_SEG - 1) / ING_SEG);
    const size_t segb = ((size_t)nseg * 4 + 255) & ~(size_t)255;
    const size_t relb = ((size_t)nseg * ING_MAXREC * 2 + 255) & ~(size_t)255;
    rc = ing_grow(g, g->seg, 5 * segb + relb);
    if (rc) return rc;
    unsigned short *d_rel = (unsigned short *)((char *)g->seg.p + 4 * segb);
    unsigned *d_hint = (unsigned *)((char *)g->seg.p + 4 * segb + relb);
    unsigned *d_first = (unsigned *)g->seg.p, *d_exit = (unsigned *)((char *)g->seg.p + segb), *d_count = (unsigned *)((char *)g->seg.p + 2 * segb),
             *d_base = (unsigned *)((char *)g->seg.p + 3 * segb);
    if (g->pin.cap < 4 * segb + 8 * (ING_EDGES + 1)) {
        if (g->pin.p) (void)hipHostFree(g->pin.p);
        g->pin.cap = 4 * segb + 8 * (ING_EDGES + 1) + segb;
        TDT_HIP(hipHostMalloc(&g->pin.p, g->pin.cap));
    }
    unsigned *h_first = (unsigned *)g->pin.p, *h_exit = (unsigned *)((char *)g->pin.p + segb), *h_count = (unsigned *)((char *)g->pin.p + 2 * segb),
             *h_base = (unsigned *)((char *)g->pin.p + 3 * segb);
    if (g->tev[0]) (void)hipEventRecord(g->tev[0], st);
    hipLaunchKernelGGL(bam_find_first, dim3((nseg + 3) / 4), dim3(256), 0, st, d_out, (long long)T, unknown_start ? -1ll : (long long)skip,
                       (long long)limit, g->n_ref, nseg, d_hint);
    TDT_CHECK_LAUNCH();
    hipLaunchKernelGGL(bam_find_records, dim3((nseg + 63) / 64), dim3(64), 0, st, d_out, (long long)T, unknown_start ? -1ll : (long long)skip,
                       (long long)limit, g->n_ref, nseg, d_first, d_exit, d_count, d_rel, d_hint);
    TDT_CHECK_LAUNCH();
    if (g->tev[1]) (void)hipEventRecord(g->tev[1], st);
    g->t_have_find = g->tev[0] && g->tev[1];
    TDT_HIP(hipMemcpyAsync(h_first, d_first, 3 * segb, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    const double t_chain0 = ing_now_ms();
    for (int s = 0; s < nseg; s++) h_base[s] = ING_NONE;
    size_t cur = skip, n = 0;
    bool confirmed = getenv("TIDDIT_INGEST_HOST_CHASE") == nullptr;
    if (unknown_start) {
        // sharded read: the first record is the first guess of the batch; the caller cross-checks it against the
        // neighbouring shard's chain (first_off / next_off), and every later segment must agree with the chain as usual
        cur = T;
        for (int s = 0; s < nseg && (size_t)s * ING_SEG < limit; s++)
            if (h_first[s] < ING_NONE - 1) {
                cur = h_first[s];
                break;
            }
        if (cur >= limit) {
            tdt_set_error("tdt_ingest_push_bounded: no record start found in the shard's first %zu bytes", limit);
            return TDT_E_UNSUPPORTED;
        }
        confirmed = true;                                         // there is no known start a host chase could use instead
    }
    const size_t start = cur;
    while (confirmed && cur < limit) {
        const int s = (int)(cur / ING_SEG);
        if (h_first[s] != (unsigned)cur) {
            confirmed = false;
            break;
        }
        h_base[s] = (unsigned)n;
        n += h_count[s];
        if (h_exit[s] == (unsigned)cur) break;                    // the record at `cur` is incomplete: it is the tail
        cur = h_exit[s];
        if (cur < (size_t)(s + 1) * ING_SEG && cur < limit) break;  // chain stopped inside the segment: tail reached
    }
    if (!confirmed && unknown_start) {
        tdt_set_error("tdt_ingest_push_bounded: the record chain from the guessed start %zu could not be confirmed", start);
        return TDT_E_UNSUPPORTED;
    }
    bool table_dirty = false;
    if (!confirmed) {
        // A guess disagreed with the chain (or the record sanity checks are stricter than this writer): chase the
        // block_size chain serially on the host over a copy of the batch and rebuild the segment table from it.
        g->host_chases++;
        std::vector<unsigned char> raw(T);
        TDT_HIP(hipMemcpyAsync(raw.data(), d_out, T, hipMemcpyDeviceToHost, st));
        TDT_HIP(hipStreamSynchronize(st));
        for (int s = 0; s < nseg; s++) {
            h_base[s] = ING_NONE;
            h_first[s] = ING_NONE;
            h_count[s] = 0;
        }
        cur = skip;
        n = 0;
        while (cur + 4 <= T && cur < limit) {
            uint32_t bs;
            memcpy(&bs, raw.data() + cur, 4);
            if (bs < 32) {
                tdt_set_error("tdt_ingest_push: record %zu has block_size %u < 32 (corrupt stream)", n, bs);
                return TDT_E_ARG;
            }
            if (cur + 4 + (size_t)bs > T) break;
            {   // the same consistency test as tdt_bam_decode: the field kernel walks CIGAR / aux inside [record, record + bs)
                const unsigned char *r = raw.data() + cur + 4;
                int32_t lseq;
                uint16_t ncig;
                memcpy(&lseq, r + 16, 4);
                memcpy(&ncig, r + 12, 2);
                const size_t var = 32 + (size_t)r[8] + 4 * (size_t)ncig + ((size_t)(lseq < 0 ? 0 : lseq) + 1) / 2 + (size_t)(lseq < 0 ? 0 : lseq);
                if (lseq < 0 || var > bs) {
                    tdt_set_error("tdt_ingest_push: record %zu is inconsistent (fixed + variable fields exceed block_size %u)", n, bs);
                    return TDT_E_ARG;
                }
            }
            const int s = (int)(cur / ING_SEG);
            if (h_first[s] == ING_NONE) {
                h_first[s] = (unsigned)cur;
                h_base[s] = (unsigned)n;
            }
            h_count[s]++;
            n++;
            cur += 4 + (size_t)bs;
        }
        table_dirty = true;
    }
    const size_t tail = cur < T ? cur : T;
    if (first_off) *first_off = start;
    if (bounded) {
        if (cur < limit) {                                        // the shard's last record runs past the blocks that were supplied
            tdt_set_error("tdt_ingest_push_bounded: the record at %zu is incomplete; supply more blocks after the shard's own", cur);
            return TDT_E_RANGE;
        }
        if (next_off) *next_off = cur - limit;                    // where the next shard's first record starts, from its boundary
    }
    // ---- decode the fields
    if (n) {
        const size_t N = n;
        const size_t a4 = (N * 4 + 255) & ~(size_t)255, a2 = (N * 2 + 255) & ~(size_t)255, a1 = (N + 255) & ~(size_t)255, a8 = (N * 8 + 255) & ~(size_t)255;
        rc = ing_grow(g, g->soa, 9 * a4 + a2 + a1 + 3 * a8 + 8 * (ING_EDGES + 1) + 256);
        if (rc) return rc;
        char *p = (char *)g->soa.p;
        IngestOut &O = g->O;
        O.rec_off = (uint64_t *)p; p += a8;
        O.sa_off = (int64_t *)p; p += a8;
        O.packed = (unsigned long long *)p; p += a8;
        O.tid = (int32_t *)p; p += a4;
        O.pos = (int32_t *)p; p += a4;
        O.end = (int32_t *)p; p += a4;
        O.mate_tid = (int32_t *)p; p += a4;
        O.mate_pos = (int32_t *)p; p += a4;
        O.tlen = (int32_t *)p; p += a4;
        O.l_seq = (int32_t *)p; p += a4;
        O.cigar_first = (uint32_t *)p; p += a4;
        O.cigar_last = (uint32_t *)p; p += a4;
        O.flag = (uint16_t *)p; p += a2;
        O.mapq = (uint8_t *)p; p += a1;
        unsigned *d_edges = (unsigned *)p;                         // ING_EDGES edges, the counter, ING_EDGES contig ids
        TDT_HIP(hipMemcpyAsync(d_base, h_base, (size_t)nseg * 4, hipMemcpyHostToDevice, st));
        if (table_dirty) {
            TDT_HIP(hipMemcpyAsync(d_first, h_first, (size_t)nseg * 4, hipMemcpyHostToDevice, st));
            TDT_HIP(hipMemcpyAsync(d_count, h_count, (size_t)nseg * 4, hipMemcpyHostToDevice, st));
        }
        g->t_chain_ms = ing_now_ms() - t_chain0;
        if (g->tev[2]) (void)hipEventRecord(g->tev[2], st);
        if (table_dirty)
            hipLaunchKernelGGL(bam_decode_fields_serial, dim3((nseg + 63) / 64), dim3(64), 0, st, d_out, (long long)T, nseg, d_first, d_base, d_count, O);
        else
            hipLaunchKernelGGL(bam_decode_fields, dim3((nseg + 3) / 4), dim3(256), 0, st, d_out, (long long)T, nseg, d_base, d_count, d_rel, O);
        TDT_CHECK_LAUNCH();
        if (g->tev[3]) (void)hipEventRecord(g->tev[3], st);
        g->t_have_decode = g->tev[2] && g->tev[3];
        TDT_HIP(hipMemsetAsync(d_edges + ING_EDGES, 0, 4, st));
        hipLaunchKernelGGL(bam_tid_edges, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, O.tid, N, d_edges, (unsigned)ING_EDGES, d_edges + ING_EDGES);
        TDT_CHECK_LAUNCH();
        unsigned *h_edges = (unsigned *)((char *)g->pin.p + 4 * segb);
        // (the counter first: a sorted file has a handful of runs per batch, and only that many entries of either half are copied... the
        // count is not known before the copy, so both halves travel whole: 64 KB)
        TDT_HIP(hipMemcpyAsync(h_edges, d_edges, 8 * (ING_EDGES + 1) - 4, hipMemcpyDeviceToHost, st));
        TDT_HIP(hipStreamSynchronize(st));
        const unsigned ne = h_edges[ING_EDGES];
        g->edges_overflow = ne > ING_EDGES;                         // not coordinate sorted: the caller derives runs from tid itself
        g->edge_tids.clear();
        if (!g->edges_overflow) {
            std::vector<std::pair<unsigned, int32_t>> ev(ne);       // (the kernel's slots are in arrival order)
            for (unsigned k = 0; k < ne; k++) ev[k] = {h_edges[k], (int32_t)h_edges[ING_EDGES + 1 + k]};
            std::sort(ev.begin(), ev.end());
            g->edges.resize(ne);
            g->edge_tids.resize(ne);
            for (unsigned k = 0; k < ne; k++) {
                g->edges[k] = ev[k].first;
                g->edge_tids[k] = ev[k].second;
            }
        }
    }
    g->n_records = n;
    *n_records = n;
    // ---- the partial record behind the last complete one goes to the carry buffer (the batch's raw bytes stay readable where they are);
    // the next push copies it in front of its span's output
    const size_t left = bounded ? 0 : T - tail;                   // a bounded push ends the shard: nothing is carried
    if (left) {
        if (g->carrybuf.cap < left) {
            // (the old block may still be the source of the copy this push enqueued: it leaves behind the launch stream)
            rc = ing_grow(g, g->carrybuf, left + (64u << 10));
            if (rc) return rc;
        }
        TDT_HIP(hipMemcpyAsync(g->carrybuf.p, d_out + tail, left, hipMemcpyDeviceToDevice, st));
    }
    g->carry = left;
    return TDT_OK;
}

// device pointers of the current batch, in the order of tdt_bam_decode's output arguments, then the raw record bytes
extern "C" int tdt_ingest_arrays(tdt_ingest *g, const void **out14, size_t *raw_len) {
    if (!g || !out14) {
        tdt_set_error("tdt_ingest_arrays: bad argument");
        return TDT_E_ARG;
    }
    const IngestOut &O = g->O;
    const void *p[14] = {O.tid, O.pos, O.end, O.mapq, O.flag, O.mate_tid, O.mate_pos, O.tlen, O.l_seq, O.cigar_first, O.cigar_last, O.rec_off,
                         O.sa_off, g->cur >= 0 && g->slot[g->cur].out.p ? (const char *)g->slot[g->cur].out.p + g->lead : nullptr};
    for (int i = 0; i < 14; i++) out14[i] = g->n_records || i == 13 ? p[i] : nullptr;
    if (raw_len) *raw_len = g->out_len;
    return TDT_OK;
}

// From the next push on, the batch's 8-byte coverage records are cov_bin_record records for `cov`'s bin size (what
// tdt_cov_push_binned_device_multi reads); cov == NULL, or a histogram whose bin size has no binned form, returns to the generic
// cov_pack_record.  *binned tells the caller which of the two the reader now writes.
int tdt_cov_bin_spec(tdt_cov *c, CovBinSpec *out);     // tdt_coverage.hip
extern "C" int tdt_ingest_bin_for(tdt_ingest *g, tdt_cov *cov, int *binned) {
    if (!g) {
        tdt_set_error("tdt_ingest_bin_for: bad argument");
        return TDT_E_ARG;
    }
    g->O.bin = CovBinSpec();
    if (cov) {
        int rc = tdt_cov_bin_spec(cov, &g->O.bin);
        if (rc) return rc;
        if (g->O.bin.n_contigs != g->n_ref) g->O.bin = CovBinSpec();      // not this file's contig table
    }
    if (binned) *binned = g->O.bin.z != 0;
    return TDT_OK;
}

// Keep the current batch: its inflated records and field arrays move into a handle and stay valid until tdt_ingest_release, while the
// reader goes on with fresh buffers (the partial record behind the batch is carried over).  `tiddit --sv` samples its library statistics
// from the first reads of the file and then scans the whole file for signals: the sampled batches are retained and scanned where they
// lie instead of being read and inflated a second time.
struct tdt_retained {
    tdt_ctx *ctx;
    void *out, *soa;
    size_t out_cap, soa_cap;
};
extern "C" int tdt_ingest_retain(tdt_ingest *g, tdt_retained **handle) {
    if (!g || !handle) {
        tdt_set_error("tdt_ingest_retain: bad argument");
        return TDT_E_ARG;
    }
    *handle = nullptr;
    if (g->failed) {
        tdt_set_error("tdt_ingest_retain: the stream is in an error state");
        return TDT_E_ARG;
    }
    // (nothing to copy or wait for: the partial record behind the batch is in the carry buffer, and spans inflating ahead have their own slots)
    tdt_buf none;
    tdt_buf &out = g->cur >= 0 ? g->slot[g->cur].out : none;
    tdt_retained *r = new tdt_retained{g->ctx, out.p, g->soa.p, out.cap, g->soa.cap};
    out = tdt_buf();                                 // the slot allocates afresh when a span is begun in it again
    g->lead = 0;
    g->soa = tdt_buf();
    {
        IngestOut fresh_o{};
        fresh_o.bin = g->O.bin;
        g->O = fresh_o;
    }
    g->n_records = 0;
    *handle = r;
    return TDT_OK;
}
extern "C" int tdt_ingest_release(tdt_retained *r) {
    if (!r) return TDT_OK;
    (void)hipSetDevice(r->ctx->device);
    // (hipFree used to wait for the whole device; a buffer that goes to the cache is handed to its next user as it is, so every
    // stream of the context that can have touched it is drained here)
    (void)hipStreamSynchronize(r->ctx->stream);
    (void)hipStreamSynchronize(r->ctx->copy_stream);
    ing_dev_free(r->ctx->device, r->out, r->out_cap);
    ing_dev_free(r->ctx->device, r->soa, r->soa_cap);
    delete r;
    return TDT_OK;
}

// Where the last push spent its time, in milliseconds: out[0] block table (host), out[1] host-to-device copy of the compressed span
// (on the span's inflate stream; of the prefetch on the copy stream when out[6] = 1), out[2] inflate + CRC kernels of the span (elapsed on
// its inflate stream: with spans begun ahead it overlaps the previous span's tail and the launch stream's kernels), out[3] record search
// kernels, out[4] chain check on the host, out[5] field decode kernel, out[6] the span had been prefetched, out[7] wall time of the push
// call itself (a span begun ahead has its first half outside it).  The events belong to the batch's slot and to the last push: spans
// begun ahead since then do not disturb them.
extern "C" int tdt_ingest_timing(tdt_ingest *g, double *out8) {
    if (!g || !out8) {
        tdt_set_error("tdt_ingest_timing: bad argument");
        return TDT_E_ARG;
    }
    for (int i = 0; i < 8; i++) out8[i] = 0;
    if (g->cur < 0) return TDT_OK;
    const tdt_ingest::Slot &S = g->slot[g->cur];
    auto span = [&](hipEvent_t a, hipEvent_t b) -> double {
        float ms = 0;
        if (!a || !b) return 0;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        return ms;
    };
    out8[0] = S.table_ms;
    out8[1] = S.prefetched ? span(S.hit_t0, S.hit_t1) : span(S.e0, S.e1);
    out8[2] = span(S.e1, S.e2);
    out8[3] = g->t_have_find ? span(g->tev[0], g->tev[1]) : 0;
    out8[4] = g->t_chain_ms;
    out8[5] = g->t_have_decode ? span(g->tev[2], g->tev[3]) : 0;
    out8[6] = S.prefetched ? 1 : 0;
    out8[7] = g->t_wall_ms;
    return TDT_OK;
}

// the packed coverage records of the current batch (tdt_cov_push_packed_device_multi reads them; `end` serves its escapes)
extern "C" int tdt_ingest_packed(tdt_ingest *g, const uint64_t **d_packed) {
    if (!g || !d_packed) {
        tdt_set_error("tdt_ingest_packed: bad argument");
        return TDT_E_ARG;
    }
    *d_packed = g->n_records ? (const uint64_t *)g->O.packed : nullptr;
    return TDT_OK;
}

// record indices at which the contig id changes (ascending, index 0 included)
extern "C" int tdt_ingest_edges(tdt_ingest *g, uint32_t *edges, size_t cap, size_t *n) {
    if (!g || !n || (cap && !edges)) {
        tdt_set_error("tdt_ingest_edges: bad argument");
        return TDT_E_ARG;
    }
    *n = g->edges_overflow ? (size_t)-1 : g->edges.size();
    for (size_t i = 0; i < g->edges.size() && i < cap; i++) edges[i] = g->edges[i];
    return TDT_OK;
}

// the contig id of every run of tdt_ingest_edges (same order); nothing when that call reported (size_t)-1
extern "C" int tdt_ingest_edge_tids(tdt_ingest *g, int32_t *tids, size_t cap) {
    if (!g || (cap && !tids)) {
        tdt_set_error("tdt_ingest_edge_tids: bad argument");
        return TDT_E_ARG;
    }
    for (size_t i = 0; i < g->edge_tids.size() && i < cap; i++) tids[i] = g->edge_tids[i];
    return TDT_OK;
}

extern "C" int tdt_ingest_carry(tdt_ingest *g, size_t *bytes, size_t *host_chases) {
    if (!g || !bytes) {
        tdt_set_error("tdt_ingest_carry: bad argument");
        return TDT_E_ARG;
    }
    *bytes = g->carry;
    if (host_chases) *host_chases = g->host_chases;
    return TDT_OK;
}

extern "C" int tdt_copy_to_host(tdt_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !d_src))) {
        tdt_set_error("tdt_copy_to_host: bad argument");
        return TDT_E_ARG;
    }
    if (!bytes) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    TDT_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    return TDT_OK;
}
