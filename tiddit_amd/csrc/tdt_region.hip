// Regional evidence counts per SV candidate for gfx950 — the inner loop of tiddit_variant.get_region
// (tiddit_variant.pyx:54-151), which the reference runs as one random-access BAM re-scan per candidate.
// Here the contig's packed, coordinate-sorted alignment arrays stay in HBM and ONE WAVE answers one query:
//   * the reads a region fetch would return (pos < q_end and end > q_start) lie in [lo, hi) of the sorted
//     starts: hi by binary search on q_end, lo by binary search on q_start - max_span (then `end > q_start` is
//     tested per read);
//   * lane = read, 64 reads per step, the reference's predicate chain is evaluated branch-free into seven
//     per-lane counters, reduced over the wave at the end.
// All seven outputs are order-independent sums, so the result is bit-identical to the sequential loop.
#include "tdt_common.h"

struct RegionArrays {
    const int32_t *start, *end, *mate_tid, *mate_pos, *tlen;
    const uint8_t *mapq, *has_sa;
    const uint16_t *flag;
    int n;
    int tid;
    int max_span;
    long long contig_length;
};

// Wave-cooperative 64-ary lower bounds (first i with a[i] >= v) for TWO keys in lock step: every round the 64 lanes probe 64
// evenly spaced elements of each key's bracket, a ballot counts the probes below the key and the bracket shrinks 64x —
// 5 dependent loads for 25 M reads instead of the 25 of a scalar binary search, with both chains in flight together.
__device__ __forceinline__ void rg_lower_bound2(const int32_t *__restrict__ a, int n, long long v0, long long v1, int lane, int &r0,
                                                int &r1) {
    int lo0 = 0, hi0 = n, lo1 = 0, hi1 = n;
    while (hi0 - lo0 > 64 || hi1 - lo1 > 64) {
        const int c0 = (hi0 - lo0) >> 6, c1 = (hi1 - lo1) >> 6;               // chunk; 0 = this bracket is already narrow
        const int x0 = c0 ? a[lo0 + (lane + 1) * c0 - 1] : 0, x1 = c1 ? a[lo1 + (lane + 1) * c1 - 1] : 0;
        if (c0) {
            const int k = __popcll(__ballot((long long)x0 < v0));
            hi0 = k < 64 ? lo0 + (k + 1) * c0 - 1 : hi0;
            lo0 += k * c0;
        }
        if (c1) {
            const int k = __popcll(__ballot((long long)x1 < v1));
            hi1 = k < 64 ? lo1 + (k + 1) * c1 - 1 : hi1;
            lo1 += k * c1;
        }
    }
    const bool b0 = lo0 + lane < hi0 && (long long)a[lo0 + lane] < v0;
    const bool b1 = lo1 + lane < hi1 && (long long)a[lo1 + lane] < v1;
    r0 = lo0 + __popcll(__ballot(b0));
    r1 = lo1 + __popcll(__ballot(b1));
}

__global__ __launch_bounds__(256) void region_counts(RegionArrays R, const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                                     const int32_t *__restrict__ qbp, int nq, int min_q, long long max_ins,
                                                     long long *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const long long start = qs[q], end = qe[q], bp = qbp[q];
    long long q_start = start, q_end = end + max_ins;          // :68-75
    if (q_end > R.contig_length) q_end = R.contig_length;
    if (q_start >= q_end) q_start = q_end - 10;
    int lo, hi;   // pos < q_end  <=>  i < hi;   reads before lo end at or before q_start (start + max_span <= q_start)
    rg_lower_bound2(R.start, R.n, q_start - (long long)R.max_span, q_end, lane, lo, hi);
    long long bases = 0;
    unsigned n_reads = 0, low_q = 0, n_discs = 0, n_splits = 0, cross_f = 0, cross_r = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const bool in = i0 + lane < hi;
        const int i = in ? i0 + lane : lo;                                       // all eight loads issue together, predicates after
        const long long rs = R.start[i], re = R.end[i], mpos = R.mate_pos[i];
        long long isz = R.tlen[i];
        const int mtid = R.mate_tid[i];
        const unsigned f = R.flag[i];
        const bool lowq = (int)R.mapq[i] < min_q, sa = R.has_sa[i] != 0;
        bool live = in && re > q_start;                                          // returned by the region fetch
        live = live && !(f & 0x4u);                                              // :84
        live = live && !((f & 0x8u) ? rs > end : (mpos > end && rs > end));      // :89-94
        live = live && !(f & 0x400u);                                            // :96
        const bool counted = live && !(rs > end);                                // :99-102
        n_reads += counted ? 1u : 0u;
        low_q += (counted && lowq) ? 1u : 0u;
        live = live && !lowq;                                                    // :104
        cross_r += (live && rs < bp - 20 && re > bp + 20) ? 1u : 0u;             // :114
        const bool mate_bp_read = mpos < bp - 50 && re > bp + 50;                // :117
        isz = isz < 0 ? -isz : isz;
        const bool discordant = isz > max_ins || mtid != R.tid;                  // :118
        cross_f += (live && mate_bp_read && !discordant) ? 1u : 0u;              // :120
        live = live && !(re < start || rs > end);                                // :123-126
        const long long r_start = rs < start ? start : rs, r_end = re > end ? end : re;
        bases += live ? r_end - r_start + 1 : 0;                                 // :134
        n_splits += (live && sa) ? 1u : 0u;                                      // :136
        n_discs += (live && discordant) ? 1u : 0u;                               // :139
    }
    for (int d = 32; d > 0; d >>= 1) {
        bases += __shfl_xor(bases, d);
        n_reads += __shfl_xor(n_reads, d);
        low_q += __shfl_xor(low_q, d);
        n_discs += __shfl_xor(n_discs, d);
        n_splits += __shfl_xor(n_splits, d);
        cross_f += __shfl_xor(cross_f, d);
        cross_r += __shfl_xor(cross_r, d);
    }
    if (lane == 0) {
        long long *o = out + (size_t)q * 7;
        o[0] = bases; o[1] = n_reads; o[2] = low_q; o[3] = n_discs; o[4] = n_splits; o[5] = cross_f; o[6] = cross_r;
    }
}

extern "C" int tdt_region_counts_device(tdt_ctx *ctx, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq,
                                        const uint16_t *d_flag, const int32_t *d_mate_tid, const int32_t *d_mate_pos,
                                        const int32_t *d_tlen, const uint8_t *d_has_sa, size_t n, int tid, int max_span,
                                        int64_t contig_length, const int32_t *d_q_start, const int32_t *d_q_end,
                                        const int32_t *d_q_bp, size_t nq, int min_q, int64_t max_ins, int64_t *d_out) {
    if (!ctx || n >= 0x7fffffffull || nq >= 0x7fffffffull || (nq && (!d_q_start || !d_q_end || !d_q_bp || !d_out)) ||
        (n && (!d_start || !d_end || !d_mapq || !d_flag || !d_mate_tid || !d_mate_pos || !d_tlen || !d_has_sa))) {
        tdt_set_error("tdt_region_counts_device: bad argument");
        return TDT_E_ARG;
    }
    if (nq == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    RegionArrays R{d_start, d_end, d_mate_tid, d_mate_pos, d_tlen, d_mapq, d_has_sa, d_flag, (int)n, tid, max_span, (long long)contig_length};
    const unsigned blocks = (unsigned)((nq + 3) / 4);
    hipLaunchKernelGGL(region_counts, dim3(blocks), dim3(256), 0, ctx->stream, R, d_q_start, d_q_end, d_q_bp, (int)nq, min_q,
                       (long long)max_ins, (long long *)d_out);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_region_counts(tdt_ctx *ctx, const int32_t *start, const int32_t *end, const uint8_t *mapq, const uint16_t *flag,
                                 const int32_t *mate_tid, const int32_t *mate_pos, const int32_t *tlen, const uint8_t *has_sa,
                                 size_t n, int tid, int64_t contig_length, const int32_t *q_start, const int32_t *q_end,
                                 const int32_t *q_bp, size_t nq, int min_q, int64_t max_ins, int64_t *out) {
    if (!ctx || (nq && (!q_start || !q_end || !q_bp || !out)) ||
        (n && (!start || !end || !mapq || !flag || !mate_tid || !mate_pos || !tlen || !has_sa))) {
        tdt_set_error("tdt_region_counts: bad argument");
        return TDT_E_ARG;
    }
    if (nq == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    int max_span = 1;
    for (size_t i = 0; i < n; i++) {
        if (i && start[i] < start[i - 1]) {
            tdt_set_error("tdt_region_counts: reads must be coordinate sorted");
            return TDT_E_ARG;
        }
        const long long sp = (long long)end[i] - start[i];
        if (sp > max_span) max_span = (int)sp;
    }
    const size_t N = n ? n : 1;
    const size_t a4 = (N * 4 + 255) & ~(size_t)255, a2 = (N * 2 + 255) & ~(size_t)255, a1 = (N + 255) & ~(size_t)255;
    const size_t q4 = (nq * 4 + 255) & ~(size_t)255;
    void *d = nullptr;
    int rc = tdt_scratch(ctx, 14, 5 * a4 + a2 + 2 * a1 + 3 * q4 + nq * 56 + 256, &d);
    if (rc) return rc;
    char *p = (char *)d;
    int32_t *ds = (int32_t *)p; p += a4;
    int32_t *de = (int32_t *)p; p += a4;
    int32_t *dmt = (int32_t *)p; p += a4;
    int32_t *dmp = (int32_t *)p; p += a4;
    int32_t *dtl = (int32_t *)p; p += a4;
    uint16_t *df = (uint16_t *)p; p += a2;
    uint8_t *dq = (uint8_t *)p; p += a1;
    uint8_t *dsa = (uint8_t *)p; p += a1;
    int32_t *dqs = (int32_t *)p; p += q4;
    int32_t *dqe = (int32_t *)p; p += q4;
    int32_t *dqb = (int32_t *)p; p += q4;
    int64_t *dout = (int64_t *)p;
    hipStream_t st = ctx->stream;
    if (n) {
        TDT_HIP(hipMemcpyAsync(ds, start, n * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(de, end, n * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dmt, mate_tid, n * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dmp, mate_pos, n * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dtl, tlen, n * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(df, flag, n * 2, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dq, mapq, n, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemcpyAsync(dsa, has_sa, n, hipMemcpyHostToDevice, st));
    }
    TDT_HIP(hipMemcpyAsync(dqs, q_start, nq * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dqe, q_end, nq * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dqb, q_bp, nq * 4, hipMemcpyHostToDevice, st));
    rc = tdt_region_counts_device(ctx, ds, de, dq, df, dmt, dmp, dtl, dsa, n, tid, max_span, contig_length, dqs, dqe, dqb, nq, min_q,
                                  max_ins, dout);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(out, dout, nq * 56, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    return TDT_OK;
}
