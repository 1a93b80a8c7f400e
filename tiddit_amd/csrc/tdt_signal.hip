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
