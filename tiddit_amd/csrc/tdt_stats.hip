// Library statistics of `tiddit --sv` on the device — tiddit_stats.statistics (tiddit_stats.py:5-78).
//
// The reference walks the first n_reads placed alignments one by one (:17-47: read-length list, the insert-size list of the pairs
// that pass its filters, the two orientation counters) and then asks numpy for the mean, the standard deviation and the 99.9th
// percentile of the insert sizes (:52-56).  Here the decoded field arrays of every ingest batch are already in HBM:
//   st_mark    per read: placed (tid >= 0: samfile.fetch() leaves the unplaced tail out), passes the pair filters (:25-38), outtie
//              (:42-45); per 256-read tile the number of placed reads
//   (scan)     the placed-read index of every read = reads sampled before the batch + prefix: the n_sampled counter of :20
//   st_cut     applies the cut-off (:22-23: reads up to n_reads count, the (n_reads+1)-th still leaves its length behind), sums the
//              read lengths, counts orientations, sums the insert sizes, counts the passing reads per tile
//   (scan)
//   st_write   stable compaction of the passing reads' template_length behind the ones of earlier batches: the reference's list, in order
// and at the end
//   mean       exact (an integer sum below 2^53 in float64, as numpy's float64 reduction of the ints is)
//   std        numpy's _var: d2 = (x - mean)^2 elementwise, then numpy.add.reduce in its own order (8192-element pairwise chunks folded in
//              sequence: tdt_np_sum_device, shared with the region means), / n, sqrt on the host
//   percentile the two order statistics floor((n-1) q) and the next one by a 4-pass radix select; numpy's linear interpolation on the host
// Bit-identical to the numpy calls (tests: product == the reference's own tiddit_stats.py on the e2e fixtures).
#include "tdt_common.h"

#include <algorithm>

typedef unsigned long long ull;

int tdt_scan_u32_inclusive(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum);                                   // tdt_dbscan.hip
int tdt_np_sum_device(tdt_ctx *ctx, const double *d_a, size_t n, double *d_csums, double *d_out);                   // tdt_means.hip

#define ST_THREADS 256

struct StState {                 // device-resident counters
    long long sampled;           // placed reads seen so far (n_sampled, capped at n_reads + 1 where the reference breaks)
    long long sum_len, n_len;    // read_length list: sum and count
    long long innie, outtie;
    long long n_ins, sum_ins;    // insert_size list
    long long batch_placed;      // placed reads of the batch in flight
    long long batch_pass;        // ... and its reads that joined the insert-size list
};

struct tdt_stats {
    tdt_ctx *ctx = nullptr;
    long long n_reads = 0, max_ins_len = 0;
    int min_mapq = 0;
    StState *d_state = nullptr;
    int *d_ins = nullptr;        // insert sizes in list order, capacity n_reads + 1
    size_t cap = 0;
    bool done = false;
};

__global__ __launch_bounds__(ST_THREADS) void st_mark(const int *__restrict__ tid, const int *__restrict__ pos, const int *__restrict__ mate_tid,
                                                      const int *__restrict__ mate_pos, const int *__restrict__ tlen,
                                                      const unsigned short *__restrict__ flag, const unsigned char *__restrict__ mapq, int n,
                                                      int min_mapq, long long max_ins_len, unsigned char *__restrict__ mark,
                                                      unsigned *__restrict__ tile_placed) {
    __shared__ unsigned wsum[ST_THREADS / 64];
    const int i = blockIdx.x * ST_THREADS + threadIdx.x;
    unsigned char mk = 0;
    if (i < n) {
        const int t = tid[i];
        if (t >= 0) {
            mk = 1;
            const unsigned f = flag[i];
            const bool rev = (f & 0x10u) != 0, mrev = (f & 0x20u) != 0;
            const bool ok = !(f & 0x8u) && rev != mrev && mate_tid[i] == t && (long long)tlen[i] <= max_ins_len && mate_pos[i] >= pos[i] &&
                            !(f & 0xd00u) && (int)mapq[i] >= min_mapq;                                           // :25-38
            if (ok) mk |= 2 | ((rev && !mrev) ? 4 : 0);                                                           // :42-45
        }
        mark[i] = mk;
    }
    const ull b = __ballot(mk & 1);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) tile_placed[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// tile_placed holds INCLUSIVE prefix sums here
__global__ __launch_bounds__(ST_THREADS) void st_cut(const int *__restrict__ tlen, const int *__restrict__ l_seq, int n, long long n_reads,
                                                     const unsigned *__restrict__ tile_placed, unsigned char *__restrict__ mark,
                                                     unsigned *__restrict__ tile_pass, StState *__restrict__ S) {
    __shared__ unsigned wsum[ST_THREADS / 64], wpass[ST_THREADS / 64];
    __shared__ long long red[5][ST_THREADS / 64];
    const int i = blockIdx.x * ST_THREADS + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char mk = i < n ? mark[i] : 0;
    const ull b = __ballot(mk & 1);
    if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
    __syncthreads();
    long long idx = S->sampled + (blockIdx.x ? tile_placed[blockIdx.x - 1] : 0u);
    for (int w = 0; w < wave; w++) idx += wsum[w];
    idx += __popcll(b & ((2ull << lane) - 1ull));                  // inclusive: the read's own n_sampled value (:20)
    const bool placed = mk & 1;
    const bool in_len = placed && idx <= n_reads + 1;              // read_length.append comes before the test (:19-23)
    const bool pass = placed && (mk & 2) && idx <= n_reads;
    long long v[5] = {in_len ? (long long)l_seq[i] : 0, in_len ? 1 : 0, (pass && !(mk & 4)) ? 1 : 0, (pass && (mk & 4)) ? 1 : 0,
                      pass ? (long long)tlen[i] : 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        for (int d = 32; d > 0; d >>= 1) v[k] += __shfl_xor(v[k], d);
        if (lane == 0) red[k][wave] = v[k];
    }
    const ull pb = __ballot(pass);
    if (lane == 0) wpass[wave] = (unsigned)__popcll(pb);
    if (i < n) mark[i] = pass ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_pass[blockIdx.x] = wpass[0] + wpass[1] + wpass[2] + wpass[3];
        long long t[5];
        for (int k = 0; k < 5; k++) t[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        if (t[0]) atomicAdd((ull *)&S->sum_len, (ull)t[0]);
        if (t[1]) atomicAdd((ull *)&S->n_len, (ull)t[1]);
        if (t[2]) atomicAdd((ull *)&S->innie, (ull)t[2]);
        if (t[3]) atomicAdd((ull *)&S->outtie, (ull)t[3]);
        if (t[4]) atomicAdd((ull *)&S->sum_ins, (ull)t[4]);      // (two's complement: negative template lengths add up correctly)
    }
}

__global__ __launch_bounds__(ST_THREADS) void st_write(const int *__restrict__ tlen, int n, const unsigned char *__restrict__ mark,
                                                       const unsigned *__restrict__ tile_pass, const StState *__restrict__ S, int *__restrict__ ins,
                                                       long long cap) {
    __shared__ unsigned wsum[ST_THREADS / 64];
    const int i = blockIdx.x * ST_THREADS + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool pass = i < n && mark[i];
    const ull b = __ballot(pass);
    if (lane == 0) wsum[wave] = (unsigned)__popcll(b);
    __syncthreads();
    long long o = S->n_ins + (blockIdx.x ? tile_pass[blockIdx.x - 1] : 0u);
    for (int w = 0; w < wave; w++) o += wsum[w];
    o += __popcll(b & ((1ull << lane) - 1ull));
    if (pass && o < cap) ins[o] = tlen[i];
}

__global__ void st_commit(StState *S, const unsigned *tile_placed, const unsigned *tile_pass, int ntiles, long long n_reads) {
    if (blockIdx.x || threadIdx.x) return;
    const long long placed = ntiles ? tile_placed[ntiles - 1] : 0, pass = ntiles ? tile_pass[ntiles - 1] : 0;
    S->batch_placed = placed;
    S->batch_pass = pass;
    S->sampled = std::min(S->sampled + placed, n_reads + 1);
    S->n_ins += pass;
}

extern "C" int tdt_stats_create(tdt_ctx *ctx, int64_t n_reads, int min_mapq, int64_t max_ins_len, tdt_stats **out) {
    if (!ctx || !out || n_reads < 0) {
        tdt_set_error("tdt_stats_create: bad argument");
        return TDT_E_ARG;
    }
    *out = nullptr;
    TDT_HIP(hipSetDevice(ctx->device));
    tdt_stats *s = new tdt_stats();
    s->ctx = ctx;
    s->n_reads = n_reads;
    s->min_mapq = min_mapq;
    s->max_ins_len = max_ins_len;
    s->cap = (size_t)n_reads + 1;
    if (tdt_dev_malloc((void **)&s->d_state, sizeof(StState)) != hipSuccess || tdt_dev_malloc((void **)&s->d_ins, s->cap * 4) != hipSuccess) {
        (void)hipGetLastError();
        if (s->d_state) (void)hipFree(s->d_state);
        delete s;
        tdt_set_error("tdt_stats_create: device allocation failed (%zu insert sizes)", (size_t)n_reads + 1);
        return TDT_E_NOMEM;
    }
    TDT_HIP(hipMemsetAsync(s->d_state, 0, sizeof(StState), ctx->stream));
    *out = s;
    return TDT_OK;
}

extern "C" int tdt_stats_destroy(tdt_stats *s) {
    if (!s) return TDT_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_state) (void)hipFree(s->d_state);
    if (s->d_ins) (void)hipFree(s->d_ins);
    delete s;
    return TDT_OK;
}

// one ingest batch (device field arrays, tdt_ingest_arrays order of the ones needed); *done: the sample is complete (:22-23)
extern "C" int tdt_stats_push_device(tdt_stats *s, const int32_t *d_tid, const int32_t *d_pos, const int32_t *d_mate_tid, const int32_t *d_mate_pos,
                                     const int32_t *d_tlen, const int32_t *d_l_seq, const uint16_t *d_flag, const uint8_t *d_mapq, size_t n_,
                                     int *done) {
    if (!s || !done || (n_ && (!d_tid || !d_pos || !d_mate_tid || !d_mate_pos || !d_tlen || !d_l_seq || !d_flag || !d_mapq)) || n_ >= 0x7fffff00ull) {
        tdt_set_error("tdt_stats_push_device: bad argument");
        return TDT_E_ARG;
    }
    *done = s->done ? 1 : 0;
    if (s->done || !n_) return TDT_OK;
    tdt_ctx *ctx = s->ctx;
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n = (int)n_, nt = (n + ST_THREADS - 1) / ST_THREADS;
    void *d = nullptr;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    int rc = tdt_scratch(ctx, 24, al((size_t)n) + 2 * al((size_t)nt * 4) + al(((size_t)nt / 1024 + 2) * 4) + 256, &d);
    if (rc) return rc;
    unsigned char *mark = (unsigned char *)d;
    unsigned *tp = (unsigned *)((char *)d + al((size_t)n)), *tq = (unsigned *)((char *)tp + al((size_t)nt * 4));
    unsigned *ts = (unsigned *)((char *)tq + al((size_t)nt * 4));
    hipLaunchKernelGGL(st_mark, dim3(nt), dim3(ST_THREADS), 0, st, d_tid, d_pos, d_mate_tid, d_mate_pos, d_tlen, d_flag, d_mapq, n, s->min_mapq,
                       (long long)s->max_ins_len, mark, tp);
    TDT_CHECK_LAUNCH();
    rc = tdt_scan_u32_inclusive(ctx, tp, nt, ts);
    if (rc) return rc;
    hipLaunchKernelGGL(st_cut, dim3(nt), dim3(ST_THREADS), 0, st, d_tlen, d_l_seq, n, (long long)s->n_reads, (const unsigned *)tp, mark, tq, s->d_state);
    TDT_CHECK_LAUNCH();
    rc = tdt_scan_u32_inclusive(ctx, tq, nt, ts);
    if (rc) return rc;
    hipLaunchKernelGGL(st_write, dim3(nt), dim3(ST_THREADS), 0, st, d_tlen, n, (const unsigned char *)mark, (const unsigned *)tq,
                       (const StState *)s->d_state, s->d_ins, (long long)s->cap);
    hipLaunchKernelGGL(st_commit, dim3(1), dim3(1), 0, st, s->d_state, (const unsigned *)tp, (const unsigned *)tq, nt, (long long)s->n_reads);
    TDT_CHECK_LAUNCH();
    long long sampled = 0;
    TDT_HIP(hipMemcpyAsync(&sampled, &s->d_state->sampled, 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    s->done = sampled > s->n_reads;
    *done = s->done ? 1 : 0;
    return TDT_OK;
}

// ---- the figures ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void st_sqdev(const int *__restrict__ x, long long n, double mean, double *__restrict__ d2) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) {
        const double d = (double)x[i] - mean;                      // numpy: arr - arrmean, then multiply(x, x)
        d2[i] = d * d;
    }
}

// radix select of two ranks at once on the biased 32-bit keys (order of signed values), top byte first
struct SelState {
    unsigned prefix[2];
    unsigned long long k[2];
};
__global__ __launch_bounds__(256) void sel_hist(const int *__restrict__ x, long long n, int pass, const SelState *__restrict__ S, unsigned *__restrict__ hist) {
    __shared__ unsigned h[2][256];
    h[0][threadIdx.x] = 0;
    h[1][threadIdx.x] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned himask = pass ? (~0u << (shift + 8)) : 0u;
    const unsigned p0 = S->prefix[0], p1 = S->prefix[1];
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) {
        const unsigned key = (unsigned)x[i] ^ 0x80000000u;
        const unsigned b = (key >> shift) & 255u;
        if (((key ^ p0) & himask) == 0) atomicAdd(&h[0][b], 1u);
        if (((key ^ p1) & himask) == 0) atomicAdd(&h[1][b], 1u);
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&hist[256 + threadIdx.x], h[1][threadIdx.x]);
}
__global__ __launch_bounds__(256) void sel_pick(int pass, SelState *__restrict__ S, unsigned *__restrict__ hist) {
    __shared__ unsigned long long cum[256];
    const int tid = threadIdx.x, shift = 24 - 8 * pass;
    for (int which = 0; which < 2; which++) {
        unsigned *h = hist + which * 256;
        cum[tid] = h[tid];
        __syncthreads();
        if (tid == 0) {
            unsigned long long run = 0;
            for (int b = 0; b < 256; b++) {
                const unsigned long long c = cum[b];
                cum[b] = run;
                run += c;
            }
        }
        __syncthreads();
        const unsigned long long k = S->k[which], below = cum[tid], mine = h[tid];
        __syncthreads();
        if (mine && k >= below && k < below + mine) {
            S->prefix[which] |= (unsigned)tid << shift;
            S->k[which] = k - below;
        }
        h[tid] = 0;
        __syncthreads();
    }
}

// counts[9]: sampled, sum_len, n_len, innie, outtie, n_ins, sum_ins, (2 spare).  With n_ins > 0 and mean given (float64(sum_ins) / n_ins,
// formed by the caller): *mean_sqdev = numpy's mean of (x - mean)^2; with ranks k0 <= k1 < n_ins: the two order statistics.
extern "C" int tdt_stats_counts(tdt_stats *s, int64_t *counts9) {
    if (!s || !counts9) {
        tdt_set_error("tdt_stats_counts: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(s->ctx->device));
    StState h;
    TDT_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->ctx->stream));
    TDT_HIP(hipStreamSynchronize(s->ctx->stream));
    const long long v[9] = {h.sampled, h.sum_len, h.n_len, h.innie, h.outtie, h.n_ins, h.sum_ins, h.batch_placed, h.batch_pass};
    for (int i = 0; i < 9; i++) counts9[i] = v[i];
    return TDT_OK;
}

extern "C" int tdt_stats_moments(tdt_stats *s, double mean, int64_t k0, int64_t k1, double *mean_sqdev, int32_t *order0, int32_t *order1) {
    if (!s || !mean_sqdev || !order0 || !order1) {
        tdt_set_error("tdt_stats_moments: bad argument");
        return TDT_E_ARG;
    }
    tdt_ctx *ctx = s->ctx;
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    StState h;
    TDT_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    const long long n = h.n_ins;
    if (n <= 0 || k0 < 0 || k1 < k0 || k1 >= n) {
        tdt_set_error("tdt_stats_moments: %lld insert sizes, ranks %lld / %lld", n, (long long)k0, (long long)k1);
        return TDT_E_ARG;
    }
    const size_t nch = ((size_t)n + 8191) / 8192;
    void *d = nullptr;
    int rc = tdt_scratch(ctx, 25, (size_t)n * 8 + nch * 8 + 4096, &d);
    if (rc) return rc;
    double *d2 = (double *)d, *cs = d2 + n, *dout = cs + nch;
    SelState *ss = (SelState *)(dout + 2);
    unsigned *hist = (unsigned *)(ss + 1);
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(st_sqdev, dim3(grid), dim3(256), 0, st, (const int *)s->d_ins, n, mean, d2);
    TDT_CHECK_LAUNCH();
    rc = tdt_np_sum_device(ctx, d2, (size_t)n, cs, dout);
    if (rc) return rc;
    SelState hs{};
    hs.k[0] = (unsigned long long)k0;
    hs.k[1] = (unsigned long long)k1;
    TDT_HIP(hipMemcpyAsync(ss, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemsetAsync(hist, 0, 2 * 256 * 4, st));
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(sel_hist, dim3(std::min(grid, 2048u)), dim3(256), 0, st, (const int *)s->d_ins, n, pass, (const SelState *)ss, hist);
        hipLaunchKernelGGL(sel_pick, dim3(1), dim3(256), 0, st, pass, ss, hist);
        TDT_CHECK_LAUNCH();
    }
    double total = 0;
    TDT_HIP(hipMemcpyAsync(&total, dout, 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(&hs, ss, sizeof(hs), hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    *mean_sqdev = total / (double)n;                               // ret = umr_sum(x) / rcount
    *order0 = (int32_t)(hs.prefix[0] ^ 0x80000000u);
    *order1 = (int32_t)(hs.prefix[1] ^ 0x80000000u);
    return TDT_OK;
}
