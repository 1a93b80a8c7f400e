// Region means of the coverage bins for SV candidates — tiddit_variant.pyx:265-283 (avg_a / avg_b: numpy.average of the 50-bp
// bins of [start/50, end/50]) and :307-315 (covM: numpy.average of the bins between the breakpoints whose gc != -1) — as one
// batched launch over the bins cov_finalize left in HBM, instead of one numpy call per candidate on the host.
//
// Exactness: numpy.average of a contiguous float64 slice is numpy.add.reduce / n, and the reduction is numpy's PAIRWISE
// summation fed in buffer-size chunks of 8192 elements (oracle/tiddit_oracle.c: orc_np_mean spells the order out and is
// checked against numpy itself).  A float64 sum depends on that order, so the kernel reproduces it: one wavefront per segment;
// inside an 8192-chunk the leaves of the halving tree (<= 128 elements each) are summed by groups of eight lanes — lane j of a
// group carries numpy's j-th interleaved accumulator — eight leaves at a time, combined in numpy's fixed order
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by three exchange steps (a+b == b+a bit for bit), the tail elements added in sequence;
// lane 0 then folds the leaf sums along the halving tree and the chunk sums in sequence.  Masked segments are first compacted
// in order (ballot prefix), which is what boolean indexing does.
#include "tdt_common.h"

#define MN_LEAF 128
#define MN_CHUNK 8192
#define MN_MAXLEAVES (MN_CHUNK / 64 + 2)     // a leaf holds more than 64 elements unless the chunk is small

struct MnLeaf {
    unsigned short start, len;
};

// the leaves of numpy's halving tree over n <= 8192 elements, in order (lane 0)
__device__ int mn_leaves(int n, MnLeaf *out) {
    int stack_s[16], stack_n[16], sp = 0, k = 0;
    stack_s[0] = 0;
    stack_n[0] = n;
    sp = 1;
    while (sp) {
        sp--;
        const int s = stack_s[sp], m = stack_n[sp];
        if (m <= MN_LEAF) {
            out[k].start = (unsigned short)s;
            out[k].len = (unsigned short)m;
            k++;
        } else {
            int n2 = m / 2;
            n2 -= n2 % 8;
            stack_s[sp] = s + n2;            // right half is visited after the left one
            stack_n[sp] = m - n2;
            sp++;
            stack_s[sp] = s;
            stack_n[sp] = n2;
            sp++;
        }
    }
    return k;
}

// value of the tree over the leaf sums (lane 0): pairwise(n) = pairwise(n2) + pairwise(n - n2)
__device__ double mn_fold(int n, const double *leaf, int &next) {
    // iterative post-order evaluation
    int st_n[16], st_phase[16], sp = 0;
    double st_left[16], ret = 0.0;
    st_n[0] = n;
    st_phase[0] = 0;
    sp = 1;
    while (sp) {
        const int m = st_n[sp - 1];
        if (m <= MN_LEAF) {
            ret = leaf[next++];
            sp--;
            continue;
        }
        int n2 = m / 2;
        n2 -= n2 % 8;
        if (st_phase[sp - 1] == 0) {          // descend left
            st_phase[sp - 1] = 1;
            st_n[sp] = n2;
            st_phase[sp] = 0;
            sp++;
        } else if (st_phase[sp - 1] == 1) {   // left done: keep it, descend right
            st_left[sp - 1] = ret;
            st_phase[sp - 1] = 2;
            st_n[sp] = m - n2;
            st_phase[sp] = 0;
            sp++;
        } else {
            ret = st_left[sp - 1] + ret;
            sp--;
        }
    }
    return ret;
}

__global__ __launch_bounds__(64) void seg_means(const double *__restrict__ cov, const signed char *__restrict__ gc,
                                                 const long long *__restrict__ lo, const long long *__restrict__ hi,
                                                 const unsigned char *__restrict__ masked, const long long *__restrict__ scratch_off,
                                                 double *__restrict__ scratch, int nq, double *__restrict__ mean, long long *__restrict__ count) {
    __shared__ MnLeaf leaves[MN_MAXLEAVES];
    __shared__ double lsum[MN_MAXLEAVES];
    __shared__ int s_nleaf;
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    long long n = hi[q] - lo[q];
    if (n < 0) n = 0;
    const double *a = cov + lo[q];
    if (masked[q]) {                          // a[gc > -1]: order-preserving compaction into the scratch area
        double *dst = scratch + scratch_off[q];
        const signed char *g = gc + lo[q];
        long long base = 0;
        for (long long i0 = 0; i0 < n; i0 += 64) {
            const long long i = i0 + lane;
            const bool keep = i < n && g[i] > -1;
            const unsigned long long m = __ballot(keep);
            if (keep) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = a[i];
            base += __popcll(m);
        }
        __threadfence();                      // the wave reads its own stores back below
        a = dst;
        n = base;
    }
    if (lane == 0) count[q] = n;
    double total = 0.0;
    for (long long o = 0; o < n; o += MN_CHUNK) {
        const int len = (int)(n - o < MN_CHUNK ? n - o : MN_CHUNK);
        const double *c = a + o;
        double csum = 0.0;
        if (len < 8) {
            if (lane == 0)
                for (int i = 0; i < len; i++) csum += c[i];
        } else {
            if (lane == 0) s_nleaf = mn_leaves(len, leaves);
            __syncthreads();
            const int nl = s_nleaf;
            const int grp = lane >> 3, j = lane & 7;
            for (int l0 = 0; l0 < nl; l0 += 8) {
                const int l = l0 + grp;
                double r = 0.0;
                int s = 0, m = 0;
                if (l < nl) {
                    s = leaves[l].start;
                    m = leaves[l].len;             // m >= 8 here: only a whole chunk below 8 elements has a shorter leaf
                    r = c[s + j];
                    for (int i = 8; i < m - (m % 8); i += 8) r += c[s + i + j];
                }
                // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) inside every group of eight lanes
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                if (l < nl && j == 0) {
                    for (int i = m - (m % 8); i < m; i++) r += c[s + i];
                    lsum[l] = r;
                }
            }
            __syncthreads();
            if (lane == 0) {
                int next = 0;
                csum = mn_fold(len, lsum, next);
            }
            __syncthreads();
        }
        if (lane == 0) total += csum;
    }
    if (lane == 0) mean[q] = n ? total / (double)n : __longlong_as_double(0x7ff8000000000000ll);   // numpy.average([]) is nan
}

// numpy.add.reduce of a contiguous float64 array of any length: the chunk sums (numpy's pairwise sum of each 8192-element buffer) are
// independent — one wavefront each — and a single thread then adds them in sequence, which is the order numpy folds them in.
__global__ __launch_bounds__(64) void np_chunk_sums(const double *__restrict__ a, long long n, double *__restrict__ csums) {
    __shared__ MnLeaf leaves[MN_MAXLEAVES];
    __shared__ double lsum[MN_MAXLEAVES];
    __shared__ int s_nleaf;
    const int lane = threadIdx.x;
    const long long o = (long long)blockIdx.x * MN_CHUNK;
    if (o >= n) return;
    const int len = (int)(n - o < MN_CHUNK ? n - o : MN_CHUNK);
    const double *c = a + o;
    double csum = 0.0;
    if (len < 8) {
        if (lane == 0)
            for (int i = 0; i < len; i++) csum += c[i];
    } else {
        if (lane == 0) s_nleaf = mn_leaves(len, leaves);
        __syncthreads();
        const int nl = s_nleaf;
        const int grp = lane >> 3, j = lane & 7;
        for (int l0 = 0; l0 < nl; l0 += 8) {
            const int l = l0 + grp;
            double r = 0.0;
            int s = 0, m = 0;
            if (l < nl) {
                s = leaves[l].start;
                m = leaves[l].len;
                r = c[s + j];
                for (int i = 8; i < m - (m % 8); i += 8) r += c[s + i + j];
            }
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            if (l < nl && j == 0) {
                for (int i = m - (m % 8); i < m; i++) r += c[s + i];
                lsum[l] = r;
            }
        }
        __syncthreads();
        if (lane == 0) {
            int next = 0;
            csum = mn_fold(len, lsum, next);
        }
    }
    if (lane == 0) csums[blockIdx.x] = csum;
}

__global__ void np_fold_chunks(const double *__restrict__ csums, long long nchunks, double *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    double t = 0.0;
    for (long long i = 0; i < nchunks; i++) t += csums[i];
    *out = t;
}

// *d_out (device) = numpy.add.reduce(d_a[0:n]) bit for bit; d_csums: scratch of ceil(n / 8192) doubles.  (C++ linkage: tdt_stats.hip)
int tdt_np_sum_device(tdt_ctx *ctx, const double *d_a, size_t n, double *d_csums, double *d_out) {
    const long long nch = (long long)((n + MN_CHUNK - 1) / MN_CHUNK);
    if (nch) hipLaunchKernelGGL(np_chunk_sums, dim3((unsigned)nch), dim3(64), 0, ctx->stream, d_a, (long long)n, d_csums);
    hipLaunchKernelGGL(np_fold_chunks, dim3(1), dim3(1), 0, ctx->stream, (const double *)d_csums, nch, d_out);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

static int means_launch(tdt_ctx *ctx, const double *d_cov, const int8_t *d_gc, const int64_t *seg_lo, const int64_t *seg_hi,
                        const uint8_t *masked, size_t nq, double *d_mean, int64_t *d_count) {
    hipStream_t st = ctx->stream;
    // query table + offsets of the compaction areas
    std::vector<long long> tab(3 * nq);
    long long tot = 0;
    for (size_t q = 0; q < nq; q++) {
        tab[q] = seg_lo[q];
        tab[nq + q] = seg_hi[q];
        tab[2 * nq + q] = tot;
        if (masked[q] && seg_hi[q] > seg_lo[q]) tot += seg_hi[q] - seg_lo[q];
    }
    void *d_tab = nullptr, *d_scr = nullptr;
    int rc = tdt_scratch(ctx, 22, 3 * nq * 8 + nq + 64, &d_tab);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 23, (size_t)(tot > 0 ? tot : 1) * 8, &d_scr);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_tab, tab.data(), 3 * nq * 8, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync((char *)d_tab + 3 * nq * 8, masked, nq, hipMemcpyHostToDevice, st));
    TDT_HIP(hipStreamSynchronize(st));        // `tab` is a stack object
    const long long *d_lo = (const long long *)d_tab;
    hipLaunchKernelGGL(seg_means, dim3((unsigned)nq), dim3(64), 0, st, d_cov, (const signed char *)d_gc, d_lo, d_lo + nq,
                       (const unsigned char *)((char *)d_tab + 3 * nq * 8), d_lo + 2 * nq, (double *)d_scr, (int)nq, d_mean, (long long *)d_count);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_segment_means_device(tdt_ctx *ctx, const double *d_cov, const int8_t *d_gc, const int64_t *seg_lo, const int64_t *seg_hi,
                                        const uint8_t *masked, size_t nq, double *d_mean, int64_t *d_count) {
    if (!ctx || !d_cov || !seg_lo || !seg_hi || !masked || !d_mean || !d_count || nq >= 0x7fffffffull) {
        tdt_set_error("tdt_segment_means_device: bad argument");
        return TDT_E_ARG;
    }
    if (!nq) return TDT_OK;
    for (size_t q = 0; q < nq; q++)
        if (masked[q] && !d_gc) {
            tdt_set_error("tdt_segment_means_device: a masked segment needs the gc array");
            return TDT_E_ARG;
        }
    TDT_HIP(hipSetDevice(ctx->device));
    return means_launch(ctx, d_cov, d_gc, seg_lo, seg_hi, masked, nq, d_mean, d_count);
}

extern "C" int tdt_segment_means(tdt_ctx *ctx, const double *cov, const int8_t *gc, int64_t total, const int64_t *seg_lo, const int64_t *seg_hi,
                                 const uint8_t *masked, size_t nq, double *mean, int64_t *count) {
    if (!ctx || (total && !cov) || !seg_lo || !seg_hi || !masked || !mean || !count || total < 0 || nq >= 0x7fffffffull) {
        tdt_set_error("tdt_segment_means: bad argument");
        return TDT_E_ARG;
    }
    if (!nq) return TDT_OK;
    for (size_t q = 0; q < nq; q++)
        if (seg_lo[q] < 0 || seg_hi[q] > total || (masked[q] && !gc)) {
            tdt_set_error("tdt_segment_means: segment %zu outside [0, %lld] (or masked without gc)", q, (long long)total);
            return TDT_E_ARG;
        }
    TDT_HIP(hipSetDevice(ctx->device));
    void *d = nullptr;
    const size_t T = (size_t)(total > 0 ? total : 1);
    int rc = tdt_scratch(ctx, 19, T * 9 + nq * 16 + 256, &d);
    if (rc) return rc;
    double *d_cov = (double *)d;
    int8_t *d_gc = (int8_t *)((char *)d + T * 8);
    char *p = (char *)d + ((T * 9 + 255) & ~(size_t)255);
    double *d_mean = (double *)p;
    int64_t *d_count = (int64_t *)(p + nq * 8);
    hipStream_t st = ctx->stream;
    if (total) TDT_HIP(hipMemcpyAsync(d_cov, cov, T * 8, hipMemcpyHostToDevice, st));
    if (total && gc) TDT_HIP(hipMemcpyAsync(d_gc, gc, T, hipMemcpyHostToDevice, st));
    rc = means_launch(ctx, d_cov, gc ? d_gc : nullptr, seg_lo, seg_hi, masked, nq, d_mean, d_count);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(mean, d_mean, nq * 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(count, d_count, nq * 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    return TDT_OK;
}
