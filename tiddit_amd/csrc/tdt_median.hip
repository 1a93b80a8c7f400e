// Masked per-contig medians of the coverage bins for gfx950 — the consumer of the two histograms.
//
// Replaces the Python loop + numpy.median of determine_ploidy (tiddit_coverage_analysis.pyx:14-27): for every
// contig (segment) the median of { cov[i] : cov[i] > 0 and gc[i] != -1 }, and the same over all contigs.
// Positive doubles order like their bit patterns, so the k-th smallest is found by an 8-pass radix SELECT on
// the 64-bit patterns (top byte first): one histogram pass over the candidates that still match the prefix, one
// tiny pick step per segment — no sort, nothing leaves the device but the two middle order statistics per
// segment (numpy's median of an even count is the mean of the two middle values; the host forms that mean with
// numpy so the result is bit-identical).
#include "tdt_common.h"

#define MD_THREADS 256
#define MD_ITEMS 8

typedef unsigned long long ull;

struct MedState {   // per segment, two order statistics (lower and upper middle)
    ull prefix[2];
    ull k[2];       // rank still to find among the candidates matching prefix
    ull count;      // number of selected values
    ull pad[3];
};

// hist[(seg*2 + which)*256 + byte]
__global__ __launch_bounds__(MD_THREADS) void med_hist(const double *__restrict__ cov, const signed char *__restrict__ gc,
                                                       const long long *__restrict__ seg_off, int nseg, int pass,
                                                       const MedState *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ unsigned h[2][256];
    const int tid = threadIdx.x;
    const int seg = blockIdx.y;
    const long long lo = seg_off[2 * seg], hi = seg_off[2 * seg + 1];
    const long long base = lo + (long long)blockIdx.x * (MD_THREADS * MD_ITEMS);
    if (base >= hi) return;
    h[0][tid] = 0;
    h[1][tid] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const ull himask = pass ? (~0ull << (shift + 8)) : 0ull;
    const ull p0 = state[seg].prefix[0], p1 = state[seg].prefix[1];
#pragma unroll
    for (int j = 0; j < MD_ITEMS; j++) {
        const long long i = base + (long long)j * MD_THREADS + tid;
        if (i < hi) {
            const double c = cov[i];
            if (c > 0 && gc[i] != -1) {   // `coverage > 0 and gc != -1`  (tiddit_coverage_analysis.pyx:17)
                const ull key = (ull)__double_as_longlong(c);
                const unsigned b = (unsigned)(key >> shift) & 255u;
                if (((key ^ p0) & himask) == 0) atomicAdd(&h[0][b], 1u);
                if (((key ^ p1) & himask) == 0) atomicAdd(&h[1][b], 1u);
            }
        }
    }
    __syncthreads();
    if (h[0][tid]) atomicAdd(&hist[(size_t)(seg * 2) * 256 + tid], h[0][tid]);
    if (h[1][tid]) atomicAdd(&hist[(size_t)(seg * 2 + 1) * 256 + tid], h[1][tid]);
}

// one workgroup per segment: find the byte bucket holding rank k, extend the prefix, clear the histogram
__global__ __launch_bounds__(256) void med_pick(int pass, MedState *__restrict__ state, unsigned *__restrict__ hist) {
    __shared__ ull cum[256];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int shift = 56 - 8 * pass;
    for (int which = 0; which < 2; which++) {
        unsigned *h = hist + (size_t)(seg * 2 + which) * 256;
        cum[tid] = h[tid];
        __syncthreads();
        if (tid == 0) {
            ull run = 0;
            for (int b = 0; b < 256; b++) {   // 256 serial steps, 16 times per call: negligible
                const ull c = cum[b];
                cum[b] = run;     // exclusive
                run += c;
            }
            if (pass == 0) {
                if (which == 0) state[seg].count = run;
                const ull n = run;
                state[seg].k[which] = n ? (which == 0 ? (n - 1) / 2 : n / 2) : 0;   // lower / upper middle
            }
        }
        __syncthreads();
        const ull k = state[seg].k[which];
        const ull below = cum[tid];
        const ull mine = h[tid];
        __syncthreads();
        if (mine && k >= below && k < below + mine) {   // exactly one bucket
            state[seg].prefix[which] |= (ull)tid << shift;
            state[seg].k[which] = k - below;
        }
        h[tid] = 0;
        __syncthreads();
    }
}

// the selection itself, data staged by `stage(dcov, dgc, stream)`: n elements, nseg segments [seg_off[2s], seg_off[2s+1])
template <class Stage>
static int med_run(tdt_ctx *ctx, size_t n, const int64_t *seg_off, int nseg, long long longest, Stage stage, double *lower, double *upper, int64_t *count) {
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *d = nullptr;
    const size_t a8 = (n * 8 + 255) & ~(size_t)255, a1 = (n + 255) & ~(size_t)255;
    const size_t so = ((size_t)nseg * 16 + 255) & ~(size_t)255, ss = ((size_t)nseg * sizeof(MedState) + 255) & ~(size_t)255;
    const size_t sh = (size_t)nseg * 2 * 256 * 4;
    int rc = tdt_scratch(ctx, 13, a8 + a1 + so + ss + sh + 256, &d);
    if (rc) return rc;
    char *p = (char *)d;
    double *dcov = (double *)p; p += a8;
    signed char *dgc = (signed char *)p; p += a1;
    long long *doff = (long long *)p; p += so;
    MedState *dstate = (MedState *)p; p += ss;
    unsigned *dhist = (unsigned *)p;
    if (n) {
        rc = stage(dcov, dgc, st);
        if (rc) return rc;
    }
    TDT_HIP(hipMemcpyAsync(doff, seg_off, (size_t)nseg * 16, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemsetAsync(dstate, 0, (size_t)nseg * sizeof(MedState), st));
    TDT_HIP(hipMemsetAsync(dhist, 0, sh, st));
    const unsigned gx = (unsigned)std::max<long long>(1, (longest + MD_THREADS * MD_ITEMS - 1) / (MD_THREADS * MD_ITEMS));
    for (int pass = 0; pass < 8; pass++) {
        hipLaunchKernelGGL(med_hist, dim3(gx, nseg), dim3(MD_THREADS), 0, st, (const double *)dcov, (const signed char *)dgc,
                           (const long long *)doff, nseg, pass, (const MedState *)dstate, dhist);
        hipLaunchKernelGGL(med_pick, dim3(nseg), dim3(256), 0, st, pass, dstate, dhist);
        TDT_CHECK_LAUNCH();
    }
    std::vector<MedState> hs(nseg);
    TDT_HIP(hipMemcpyAsync(hs.data(), dstate, (size_t)nseg * sizeof(MedState), hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    for (int s = 0; s < nseg; s++) {
        count[s] = (int64_t)hs[s].count;
        memcpy(&lower[s], &hs[s].prefix[0], 8);
        memcpy(&upper[s], &hs[s].prefix[1], 8);
    }
    return TDT_OK;
}

extern "C" int tdt_masked_medians(tdt_ctx *ctx, const double *cov, const int8_t *gc, const int64_t *seg_off, int nseg,
                                  double *lower, double *upper, int64_t *count) {
    if (!ctx || nseg < 1 || !seg_off || !lower || !upper || !count) {
        tdt_set_error("tdt_masked_medians: bad argument");
        return TDT_E_ARG;
    }
    long long total = 0, longest = 0;
    for (int s = 0; s < nseg; s++) {
        if (seg_off[2 * s] < 0 || seg_off[2 * s + 1] < seg_off[2 * s]) {
            tdt_set_error("tdt_masked_medians: bad segment %d", s);
            return TDT_E_ARG;
        }
        total = std::max<long long>(total, seg_off[2 * s + 1]);
        longest = std::max<long long>(longest, seg_off[2 * s + 1] - seg_off[2 * s]);
    }
    if (total && (!cov || !gc)) {
        tdt_set_error("tdt_masked_medians: null data");
        return TDT_E_ARG;
    }
    const size_t n = (size_t)total;
    return med_run(ctx, n, seg_off, nseg, longest, [&](double *dcov, signed char *dgc, hipStream_t st) -> int {
        TDT_HIP(hipMemcpyAsync(dcov, cov, n * 8, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dgc, gc, n, hipMemcpyHostToDevice, st));
        return TDT_OK;
    }, lower, upper, count);
}

// The same for data that lies in `n_parts` separate host arrays (determine_ploidy's per-contig coverage and GC arrays): the medians of
// every part and, as result n_parts, of all parts together — each part is copied to the device from where it lies (joining 60 M
// float64 bins on the host first was two thirds of the ploidy stage of a human-sized job).
extern "C" int tdt_masked_medians_parts(tdt_ctx *ctx, const double *const *cov_parts, const int8_t *const *gc_parts, const int64_t *part_len,
                                        int n_parts, double *lower, double *upper, int64_t *count) {
    if (!ctx || n_parts < 0 || (n_parts && (!cov_parts || !gc_parts || !part_len)) || !lower || !upper || !count) {
        tdt_set_error("tdt_masked_medians_parts: bad argument");
        return TDT_E_ARG;
    }
    std::vector<int64_t> seg((size_t)(n_parts + 1) * 2);
    long long o = 0, longest = 0;
    for (int s = 0; s < n_parts; s++) {
        if (part_len[s] < 0 || (part_len[s] && (!cov_parts[s] || !gc_parts[s]))) {
            tdt_set_error("tdt_masked_medians_parts: bad part %d", s);
            return TDT_E_ARG;
        }
        seg[2 * (size_t)s] = o;
        seg[2 * (size_t)s + 1] = o + part_len[s];
        o += part_len[s];
    }
    seg[2 * (size_t)n_parts] = 0;
    seg[2 * (size_t)n_parts + 1] = o;
    longest = o;
    const size_t n = (size_t)o;
    return med_run(ctx, n, seg.data(), n_parts + 1, longest, [&](double *dcov, signed char *dgc, hipStream_t st) -> int {
        for (int s = 0; s < n_parts; s++) {
            if (!part_len[s]) continue;
            TDT_HIP(hipMemcpyAsync(dcov + seg[2 * (size_t)s], cov_parts[s], (size_t)part_len[s] * 8, hipMemcpyHostToDevice, st));
            TDT_HIP(hipMemcpyAsync(dgc + seg[2 * (size_t)s], gc_parts[s], (size_t)part_len[s], hipMemcpyHostToDevice, st));
        }
        return TDT_OK;
    }, lower, upper, count);
}
