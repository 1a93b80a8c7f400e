// Discordant-pair candidate selection for gfx950: the per-read predicate of tiddit_signal.worker
// (tiddit_signal.pyx:171-211) evaluated on the packed flag/mapq/tid/mate/tlen arrays, and the surviving
// read indices compacted in stream order.
//   sig_masks    lane = read: predicate -> 64-bit ballot word, tile count (4096 reads per workgroup)
//   sig_compact  tile prefix = sum of the earlier tiles' counts (reduced in the workgroup), then wavefront
//                compaction: a read's slot = prefix + set bits below its lane (mbcnt) — order preserving.
#include "tdt_common.h"

#define SG_THREADS 256
#define SG_WORDS 64
#define SG_TILE (SG_WORDS * 64)

typedef unsigned long long ull;

struct SigParams {
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *tid;
    const int32_t *mate_tid;
    const int32_t *tlen;
    const uint8_t *contig_ok;
    int n_contigs;
    int n;
    int min_q;
    long long max_ins;
};

__global__ __launch_bounds__(SG_THREADS) void sig_masks(SigParams P, ull *__restrict__ MASK, unsigned *__restrict__ tcount) {
    __shared__ unsigned wc[SG_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * SG_TILE;
    unsigned cnt = 0;
    for (int s = 0; s < SG_WORDS / (SG_THREADS / 64); s++) {
        const int W = wave * (SG_WORDS / (SG_THREADS / 64)) + s;
        const int i = t0 + W * 64 + lane;
        bool keep = false;
        if (i < P.n) {
            const unsigned f = P.flag[i];
            const int t = P.tid[i], mt = P.mate_tid[i];
            long long isz = P.tlen[i];
            isz = isz < 0 ? -isz : isz;
            keep = t >= 0 && t < P.n_contigs && P.contig_ok[t]      // worker() only runs on contigs >= min_contig (:250-256)
                   && !(f & 0x404u)                                  // unmapped / duplicate (:171)
                   && !(f & 0x900u)                                  // supplementary / secondary (:184)
                   && (int)P.mapq[i] >= P.min_q                      // (:188)
                   && !(f & 0x8u) && (f & 0x1u)                      // mate mapped, paired (:204-208)
                   && mt >= 0 && (isz > P.max_ins || mt != t);       // discordant (:211)
        }
        const ull m = __ballot(keep);
        if (lane == 0) MASK[(size_t)blockIdx.x * SG_WORDS + W] = m;
        cnt += (unsigned)__popcll(m);
    }
    if (lane == 0) wc[wave] = cnt;
    __syncthreads();
    if (tid == 0) tcount[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(SG_THREADS) void sig_compact(const ull *__restrict__ MASK, const unsigned *__restrict__ tcount, int n,
                                                          unsigned *__restrict__ out_idx, unsigned long long *__restrict__ total) {
    __shared__ unsigned red[SG_THREADS / 64];
    __shared__ unsigned wbase[SG_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    // reads selected in earlier tiles
    unsigned pre = 0;
    for (int i = tid; i < tile; i += SG_THREADS) pre += tcount[i];
    for (int d = 32; d > 0; d >>= 1) pre += __shfl_xor(pre, d);
    if (lane == 0) red[wave] = pre;
    __syncthreads();
    pre = red[0] + red[1] + red[2] + red[3];
    if (wave == 0) {   // lane = word: exclusive scan of the words' popcounts
        const unsigned c = (unsigned)__popcll(MASK[(size_t)tile * SG_WORDS + lane]);
        unsigned s = c;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(s, d);
            if (lane >= d) s += t;
        }
        wbase[lane] = pre + s - c;
        if (lane == 63 && tile == (int)gridDim.x - 1) *total = (unsigned long long)pre + s;
    }
    __syncthreads();
    const int t0 = tile * SG_TILE;
    for (int s = 0; s < SG_WORDS / (SG_THREADS / 64); s++) {
        const int W = wave * (SG_WORDS / (SG_THREADS / 64)) + s;
        const ull m = MASK[(size_t)tile * SG_WORDS + W];
        if ((m >> lane) & 1ull) {
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            out_idx[wbase[W] + below] = (unsigned)(t0 + W * 64 + lane);
        }
    }
}

extern "C" int tdt_signal_select_device(tdt_ctx *ctx, const uint16_t *d_flag, const uint8_t *d_mapq, const int32_t *d_tid,
                                        const int32_t *d_mate_tid, const int32_t *d_tlen, size_t n_, const uint8_t *d_contig_ok,
                                        int n_contigs, int min_q, int64_t max_ins, uint32_t *d_out_idx, uint64_t *d_count) {
    if (!ctx || !d_count || (n_ && (!d_flag || !d_mapq || !d_tid || !d_mate_tid || !d_tlen || !d_contig_ok || !d_out_idx))) {
        tdt_set_error("tdt_signal_select_device: bad argument");
        return TDT_E_ARG;
    }
    if (n_ >= 0x7fffffffull) {
        tdt_set_error("tdt_signal_select_device: n too large");
        return TDT_E_UNSUPPORTED;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    const int n = (int)n_;
    if (n == 0) {
        TDT_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
        return TDT_OK;
    }
    const int nt = (n + SG_TILE - 1) / SG_TILE;
    void *scr = nullptr;
    int rc = tdt_scratch(ctx, 11, (size_t)nt * SG_WORDS * 8 + (size_t)nt * 4 + 64, &scr);
    if (rc) return rc;
    ull *MASK = (ull *)scr;
    unsigned *tcount = (unsigned *)(MASK + (size_t)nt * SG_WORDS);
    SigParams P{d_flag, d_mapq, d_tid, d_mate_tid, d_tlen, d_contig_ok, n_contigs, n, min_q, (long long)max_ins};
    hipLaunchKernelGGL(sig_masks, dim3(nt), dim3(SG_THREADS), 0, ctx->stream, P, MASK, tcount);
    hipLaunchKernelGGL(sig_compact, dim3(nt), dim3(SG_THREADS), 0, ctx->stream, (const ull *)MASK, (const unsigned *)tcount, n, d_out_idx,
                       (unsigned long long *)d_count);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_signal_select(tdt_ctx *ctx, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mate_tid,
                                 const int32_t *tlen, size_t n, const uint8_t *contig_ok, int n_contigs, int min_q, int64_t max_ins,
                                 uint32_t *out_idx, size_t *out_count) {
    if (!ctx || !out_count || n_contigs < 0 || (n && (!flag || !mapq || !tid || !mate_tid || !tlen || !contig_ok || !out_idx))) {
        tdt_set_error("tdt_signal_select: bad argument");
        return TDT_E_ARG;
    }
    *out_count = 0;
    if (n == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    const size_t nc = (size_t)(n_contigs ? n_contigs : 1);
    const size_t a4 = (n * 4 + 255) & ~(size_t)255, a2 = (n * 2 + 255) & ~(size_t)255, a1 = (n + 255) & ~(size_t)255;
    void *d = nullptr;
    int rc = tdt_scratch(ctx, 12, 4 * a4 + a2 + a1 + ((nc + 255) & ~(size_t)255) + 64, &d);
    if (rc) return rc;
    char *p = (char *)d;
    int32_t *dt = (int32_t *)p; p += a4;
    int32_t *dm = (int32_t *)p; p += a4;
    int32_t *dl = (int32_t *)p; p += a4;
    uint32_t *di = (uint32_t *)p; p += a4;
    uint16_t *df = (uint16_t *)p; p += a2;
    uint8_t *dq = (uint8_t *)p; p += a1;
    uint8_t *dc = (uint8_t *)p; p += (nc + 255) & ~(size_t)255;
    uint64_t *dn = (uint64_t *)p;
    hipStream_t st = ctx->stream;
    TDT_HIP(hipMemcpyAsync(dt, tid, n * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dm, mate_tid, n * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dl, tlen, n * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(df, flag, n * 2, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dq, mapq, n, hipMemcpyHostToDevice, st));
    if (n_contigs) TDT_HIP(hipMemcpyAsync(dc, contig_ok, (size_t)n_contigs, hipMemcpyHostToDevice, st));
    rc = tdt_signal_select_device(ctx, df, dq, dt, dm, dl, n, dc, n_contigs, min_q, max_ins, di, dn);
    if (rc) return rc;
    uint64_t cnt = 0;
    TDT_HIP(hipMemcpyAsync(&cnt, dn, 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (cnt) TDT_HIP(hipMemcpy(out_idx, di, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    *out_count = (size_t)cnt;
    return TDT_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The whole per-read chain of tiddit_signal.worker (tiddit_signal.pyx:171-221) on a decoded batch that is resident in HBM
// (tdt_ingest_arrays): one action byte per read —
//   bit 1 (2): clipped read for local assembly (:190-197)   bit 2 (4): carries an SA tag, SA_analysis is called (:199-202)
//   bit 3 (8): a discordant-pair row is appended (:204-221)
// — the reads with any bit set compacted in stream order, and everything the host needs to build their rows (fields, raw record
// bytes: names, CIGARs, sequences, SA strings) gathered into compact arrays, so that the host touches only those few per cent.
int tdt_scan_u32_inclusive(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum);   // tdt_dbscan.hip

struct ScanParams {
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *tid, *mate_tid, *tlen;
    const uint32_t *cig_first, *cig_last;
    const long long *sa_off;
    const uint8_t *contig_ok;
    int n_contigs, n, min_q;
    long long max_ins;
    int min_anchor_len, min_clip_len;
};

__global__ __launch_bounds__(SG_THREADS) void sig_actions(ScanParams P, uint8_t *__restrict__ act, ull *__restrict__ MASK, unsigned *__restrict__ tcount) {
    __shared__ unsigned wc[SG_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * SG_TILE;
    unsigned cnt = 0;
    for (int s = 0; s < SG_WORDS / (SG_THREADS / 64); s++) {
        const int W = wave * (SG_WORDS / (SG_THREADS / 64)) + s;
        const int i = t0 + W * 64 + lane;
        unsigned a = 0;
        if (i < P.n) {
            const unsigned f = P.flag[i];
            const int t = P.tid[i], mt = P.mate_tid[i];
            if (t >= 0 && t < P.n_contigs && P.contig_ok[t] && !(f & 0x404u)       // worker() runs on contigs >= min_contig; unmapped / duplicate (:171)
                && !(f & 0x900u) && (int)P.mapq[i] >= P.min_q) {                    // supplementary / secondary (:184), mapq (:188)
                long long isz = P.tlen[i];
                isz = isz < 0 ? -isz : isz;
                const bool same = mt == t;
                const unsigned cf = P.cig_first[i], cl = P.cig_last[i];
                if (isz < P.max_ins && same && cf != 0xffffffffu) {                  // :191-197
                    const unsigned f_op = cf & 0xfu, l_op = cl & 0xfu;
                    const long long f_len = cf >> 4, l_len = cl >> 4;
                    if ((f_op == 4 && f_len > P.min_clip_len && l_op == 0 && l_len > P.min_anchor_len) ||
                        (l_op == 4 && l_len > P.min_clip_len && f_op == 0 && f_len > P.min_anchor_len))
                        a |= 2u;
                }
                if (P.sa_off[i] >= 0) a |= 4u;                                       // :199
                if (!(f & 0x8u) && (f & 0x1u) && mt >= 0 && (isz > P.max_ins || !same)) a |= 8u;   // :204-211
            }
            act[i] = (uint8_t)a;
        }
        const ull m = __ballot(a != 0);
        if (lane == 0) MASK[(size_t)blockIdx.x * SG_WORDS + W] = m;
        cnt += (unsigned)__popcll(m);
    }
    if (lane == 0) wc[wave] = cnt;
    __syncthreads();
    if (tid == 0) tcount[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

struct ScanMeta {          // one selected read, as the host receives it
    uint32_t idx;
    int32_t tid, pos, end, mate_tid, sa_rel;     // sa_rel: offset of the SA:Z string inside the record, -1 without
    uint16_t flag;
    uint8_t action, pad;
};

__global__ __launch_bounds__(256) void sig_gather_meta(const unsigned *__restrict__ idx, int m, const uint8_t *__restrict__ act,
                                                       const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, const int32_t *__restrict__ end,
                                                       const uint16_t *__restrict__ flag, const int32_t *__restrict__ mate_tid,
                                                       const unsigned long long *__restrict__ rec_off, const long long *__restrict__ sa_off,
                                                       const uint8_t *__restrict__ raw, ScanMeta *__restrict__ meta, unsigned *__restrict__ size) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const unsigned i = idx[k];
    const unsigned long long o = rec_off[i];
    int bs;
    memcpy(&bs, raw + o, 4);                         // block_size (records are not aligned)
    ScanMeta r;
    r.idx = i;
    r.tid = tid[i];
    r.pos = pos[i];
    r.end = end[i];
    r.mate_tid = mate_tid[i];
    r.sa_rel = sa_off[i] >= 0 ? (int32_t)(sa_off[i] - (long long)o) : -1;
    r.flag = flag[i];
    r.action = act[i];
    r.pad = 0;
    meta[k] = r;
    size[k] = (unsigned)(bs + 4);
}

__global__ __launch_bounds__(64) void sig_gather_bytes(const unsigned *__restrict__ idx, int m, const unsigned long long *__restrict__ rec_off,
                                                       const unsigned *__restrict__ size_incl, const uint8_t *__restrict__ raw, uint8_t *__restrict__ out) {
    const int k = blockIdx.x;
    if (k >= m) return;
    const unsigned end = size_incl[k], beg = k ? size_incl[k - 1] : 0u;
    const uint8_t *src = raw + rec_off[idx[k]];
    for (unsigned b = threadIdx.x; b < end - beg; b += 64) out[beg + b] = src[b];
}

// (the result of a scan belongs to the context it ran on: tdt_signal_scan_result fetches it from there, from any thread)

extern "C" int tdt_signal_scan(tdt_ctx *ctx, const void *const *d_arrays14, size_t n_, const uint8_t *contig_ok, int n_contigs, int min_q,
                               int64_t max_ins, int min_anchor_len, int min_clip_len, size_t *n_sel, size_t *raw_bytes) {
    if (!ctx || !n_sel || !raw_bytes || n_contigs < 0 || (n_ && (!d_arrays14 || !contig_ok))) {
        tdt_set_error("tdt_signal_scan: bad argument");
        return TDT_E_ARG;
    }
    *n_sel = 0;
    *raw_bytes = 0;
    ctx->scan_n_sel = ctx->scan_raw_bytes = 0;
    ctx->scan_meta = ctx->scan_size = ctx->scan_bytes = nullptr;
    if (n_ == 0) return TDT_OK;
    if (n_ >= 0x7fffffffull) {
        tdt_set_error("tdt_signal_scan: n too large");
        return TDT_E_UNSUPPORTED;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n = (int)n_;
    const int nt = (n + SG_TILE - 1) / SG_TILE;
    const size_t nc = (size_t)(n_contigs ? n_contigs : 1);
    void *scr = nullptr;
    const size_t a_mask = ((size_t)nt * SG_WORDS * 8 + 255) & ~(size_t)255, a_tc = ((size_t)nt * 4 + 255) & ~(size_t)255,
                 a_act = ((size_t)n + 255) & ~(size_t)255, a_idx = ((size_t)n * 4 + 255) & ~(size_t)255, a_ok = (nc + 255) & ~(size_t)255;
    int rc = tdt_scratch(ctx, 4, a_mask + a_tc + a_act + a_idx + a_ok + 256, &scr);
    if (rc) return rc;
    char *p = (char *)scr;
    ull *MASK = (ull *)p; p += a_mask;
    unsigned *tcount = (unsigned *)p; p += a_tc;
    uint8_t *d_act = (uint8_t *)p; p += a_act;
    unsigned *d_idx = (unsigned *)p; p += a_idx;
    uint8_t *d_ok = (uint8_t *)p; p += a_ok;
    unsigned long long *d_total = (unsigned long long *)p;
    if (n_contigs) TDT_HIP(hipMemcpyAsync(d_ok, contig_ok, (size_t)n_contigs, hipMemcpyHostToDevice, st));
    ScanParams P;
    P.tid = (const int32_t *)d_arrays14[0];
    const int32_t *d_pos = (const int32_t *)d_arrays14[1], *d_end = (const int32_t *)d_arrays14[2];
    P.mapq = (const uint8_t *)d_arrays14[3];
    P.flag = (const uint16_t *)d_arrays14[4];
    P.mate_tid = (const int32_t *)d_arrays14[5];
    P.tlen = (const int32_t *)d_arrays14[7];
    P.cig_first = (const uint32_t *)d_arrays14[9];
    P.cig_last = (const uint32_t *)d_arrays14[10];
    const unsigned long long *d_rec_off = (const unsigned long long *)d_arrays14[11];
    P.sa_off = (const long long *)d_arrays14[12];
    const uint8_t *d_raw = (const uint8_t *)d_arrays14[13];
    P.contig_ok = d_ok;
    P.n_contigs = n_contigs;
    P.n = n;
    P.min_q = min_q;
    P.max_ins = (long long)max_ins;
    P.min_anchor_len = min_anchor_len;
    P.min_clip_len = min_clip_len;
    hipLaunchKernelGGL(sig_actions, dim3(nt), dim3(SG_THREADS), 0, st, P, d_act, MASK, tcount);
    hipLaunchKernelGGL(sig_compact, dim3(nt), dim3(SG_THREADS), 0, st, (const ull *)MASK, (const unsigned *)tcount, n, d_idx, d_total);
    TDT_CHECK_LAUNCH();
    unsigned long long cnt = 0;
    TDT_HIP(hipMemcpyAsync(&cnt, d_total, 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (!cnt) return TDT_OK;
    const int m = (int)cnt;
    void *scr2 = nullptr;
    const size_t a_meta = ((size_t)m * sizeof(ScanMeta) + 255) & ~(size_t)255, a_size = ((size_t)m * 4 + 255) & ~(size_t)255,
                 a_ts = ((size_t)(m / 1024 + 2) * 4 + 255) & ~(size_t)255;
    rc = tdt_scratch(ctx, 17, a_meta + a_size + a_ts, &scr2);
    if (rc) return rc;
    ScanMeta *d_meta = (ScanMeta *)scr2;
    unsigned *d_size = (unsigned *)((char *)scr2 + a_meta);
    unsigned *d_ts = (unsigned *)((char *)scr2 + a_meta + a_size);
    hipLaunchKernelGGL(sig_gather_meta, dim3((m + 255) / 256), dim3(256), 0, st, (const unsigned *)d_idx, m, (const uint8_t *)d_act, P.tid, d_pos, d_end,
                       P.flag, P.mate_tid, d_rec_off, P.sa_off, d_raw, d_meta, d_size);
    TDT_CHECK_LAUNCH();
    rc = tdt_scan_u32_inclusive(ctx, d_size, m, d_ts);
    if (rc) return rc;
    unsigned total = 0;
    TDT_HIP(hipMemcpyAsync(&total, d_size + (m - 1), 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    void *scr3 = nullptr;
    rc = tdt_scratch(ctx, 18, (size_t)total + 256, &scr3);
    if (rc) return rc;
    hipLaunchKernelGGL(sig_gather_bytes, dim3(m), dim3(64), 0, st, (const unsigned *)d_idx, m, d_rec_off, (const unsigned *)d_size, d_raw, (uint8_t *)scr3);
    TDT_CHECK_LAUNCH();
    // everything that reads the batch is enqueued: the caller may start the next span's inflate behind it (tdt_ingest_push_ahead); the
    // result copies wait for THIS point, on a stream of their own
    TDT_HIP(hipEventRecord(ctx->ev[1], st));
    ctx->scan_n_sel = (size_t)m;
    ctx->scan_raw_bytes = total;
    ctx->scan_meta = d_meta;
    ctx->scan_size = d_size;
    ctx->scan_bytes = scr3;
    *n_sel = (size_t)m;
    *raw_bytes = total;
    return TDT_OK;
}

extern "C" int tdt_signal_scan_result(tdt_ctx *ctx, void *meta24, uint32_t *raw_end, uint8_t *raw) {
    if (!ctx || (ctx->scan_n_sel && (!meta24 || !raw_end || !raw))) {
        tdt_set_error("tdt_signal_scan_result: bad argument");
        return TDT_E_ARG;
    }
    if (!ctx->scan_n_sel) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    // the copies run on the context's return stream behind the scan's last kernel — not on the launch stream, where the next span's
    // inflate kernel (enqueued ahead by the caller, ~12 ms) may already stand between that kernel and them
    hipStream_t st = ctx->back_stream;
    static_assert(sizeof(ScanMeta) == 28, "ScanMeta layout");
    TDT_HIP(hipStreamWaitEvent(st, ctx->ev[1], 0));
    TDT_HIP(hipMemcpyAsync(meta24, ctx->scan_meta, ctx->scan_n_sel * sizeof(ScanMeta), hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(raw_end, ctx->scan_size, ctx->scan_n_sel * 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(raw, ctx->scan_bytes, ctx->scan_raw_bytes, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    return TDT_OK;
}
