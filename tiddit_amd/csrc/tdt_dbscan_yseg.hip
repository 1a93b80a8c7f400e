// DBSCAN.y_coordinate_clustering (DBSCAN.py:66-123) on ARBITRARY label arrays, for gfx950.
//
// The reference visits `set(clusters)` and, for every value, takes the members by VALUE (`clusters == cluster`): members need
// not be contiguous, labels need not be dense, a cluster may have any size.  Per visited cluster it sorts the members by y
// (stable: ties keep index order), runs the sliding-window state machine with m-1 following points (:92-110), and relabels
// (:112-122): sub-run 1 keeps the cluster's value, sub-run s > 1 becomes s + cluster_id - 1, the rest -1; cluster_id grows by
// the number of extra sub-runs.  As long as no produced id equals a value that is still to be visited, the clusters are
// independent and the pass is data parallel over SEGMENTS (one per visited value, numbered in visiting order by the caller):
//   sort (segment, y) stably   -> members of a segment contiguous, ordered as the reference's y_coordinates
//   p[k]     = k+m-1 is still in k's segment and y[k+m-1] - y[k] < eps        (sorted: the window's maximum is its last member)
//   start[k] = p[k] and not p[k-1] (inside the segment)
//   P, S     = inclusive scans of p and start over the whole array
//   labelled(k) = some p in [max(segstart, k-m+1), k];  sub(k) = S[k] - S[segstart-1]
//   extras(seg) = max(sub-runs - 1, 0), exclusive scan in visiting order -> id base of every segment
// The caller (tiddit_amd/DBSCAN.py) supplies the visiting order — Python's own set() — and replays clusters one at a time
// through this same entry point (one segment per call) when ids could collide with unvisited values.
#include "tdt_common.h"

#include <algorithm>
#include <cmath>

typedef unsigned long long ull;

int tdt_radix_sort_pairs(tdt_ctx *ctx, ull *keys, unsigned *vals, ull *keys_tmp, unsigned *vals_tmp, size_t n, ull bitmask, ull **out_keys,
                         unsigned **out_vals);                                             // tdt_sort.hip
int tdt_scan_u32_inclusive(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum);        // tdt_dbscan.hip

#define YS_THREADS 256

__global__ __launch_bounds__(YS_THREADS) void ys_keys(const unsigned *__restrict__ y, const int *__restrict__ seg, int n, ull *__restrict__ key,
                                                       unsigned *__restrict__ val) {
    const int i = blockIdx.x * YS_THREADS + threadIdx.x;
    if (i >= n) return;
    key[i] = ((ull)(unsigned)seg[i] << 32) | y[i];
    val[i] = (unsigned)i;
}

// window predicate, run starts, segment bounds
__global__ __launch_bounds__(YS_THREADS) void ys_flags(const ull *__restrict__ key, int n, int m, ull eps, unsigned *__restrict__ P,
                                                        unsigned *__restrict__ S, int *__restrict__ seg_lo, int *__restrict__ seg_hi) {
    const int k = blockIdx.x * YS_THREADS + threadIdx.x;
    if (k >= n) return;
    const ull me = key[k];
    const unsigned s = (unsigned)(me >> 32);
    auto pred = [&](int j, ull kj) -> bool {
        if (j + m - 1 >= n) return false;
        const ull far = key[j + m - 1];
        return (unsigned)(far >> 32) == (unsigned)(kj >> 32) && (ull)((unsigned)far - (unsigned)kj) < eps;
    };
    const bool p = pred(k, me);
    bool prev = false;
    const bool first = k == 0 || (unsigned)(key[k - 1] >> 32) != s;
    if (!first) prev = pred(k - 1, key[k - 1]);
    P[k] = p ? 1u : 0u;
    S[k] = (p && !prev) ? 1u : 0u;
    if (first) seg_lo[s] = k;
    if (k == n - 1 || (unsigned)(key[k + 1] >> 32) != s) seg_hi[s] = k + 1;
}

// extra sub-runs of every segment (DBSCAN.py:121-122)
__global__ __launch_bounds__(YS_THREADS) void ys_extras(const unsigned *__restrict__ S, const int *__restrict__ seg_lo, const int *__restrict__ seg_hi,
                                                         int nseg, unsigned *__restrict__ extras) {
    const int s = blockIdx.x * YS_THREADS + threadIdx.x;
    if (s >= nseg) return;
    const int lo = seg_lo[s], hi = seg_hi[s];
    unsigned runs = 0;
    if (hi > lo) runs = S[hi - 1] - (lo > 0 ? S[lo - 1] : 0u);
    extras[s] = runs > 1 ? runs - 1 : 0u;
}

__global__ __launch_bounds__(YS_THREADS) void ys_final(const ull *__restrict__ key, const unsigned *__restrict__ val, int n, int m,
                                                        const unsigned *__restrict__ P, const unsigned *__restrict__ S,
                                                        const int *__restrict__ seg_lo, const unsigned *__restrict__ extras_incl,
                                                        const unsigned *__restrict__ extras, int nseg, const double *__restrict__ keep,
                                                        long long cluster_id, double *__restrict__ out, long long *__restrict__ last_id) {
    const int k = blockIdx.x * YS_THREADS + threadIdx.x;
    if (k == 0 && last_id) *last_id = cluster_id + (nseg ? (long long)extras_incl[nseg - 1] : 0ll);
    if (k >= n) return;
    const unsigned s = (unsigned)(key[k] >> 32);
    const int lo = seg_lo[s];
    const int a = max(lo, k - m + 1);
    const unsigned pa = a > 0 ? P[a - 1] : 0u;
    const unsigned i = val[k];
    double r = -1.0;
    if (P[k] - pa > 0u) {
        const unsigned sub = S[k] - (lo > 0 ? S[lo - 1] : 0u);
        r = sub == 1u ? keep[i] : (double)(cluster_id + (long long)(extras_incl[s] - extras[s]) + (long long)sub - 1ll);
    }
    out[i] = r;
}

extern "C" int tdt_dbscan_y_segments(tdt_ctx *ctx, const int64_t *y, const int32_t *seg, const double *keep, size_t n_, int nseg, double eps,
                                     int m, int64_t cluster_id, double *out, int64_t *last_id) {
    if (!ctx || nseg < 0 || (n_ && (!y || !seg || !keep || !out))) {
        tdt_set_error("tdt_dbscan_y_segments: bad argument");
        return TDT_E_ARG;
    }
    if (m < 2) {
        tdt_set_error("tdt_dbscan_y_segments: m must be >= 2");
        return TDT_E_ARG;
    }
    if (last_id) *last_id = cluster_id;
    if (n_ == 0) return TDT_OK;
    if (n_ >= 0x7fff0000ull) {
        tdt_set_error("tdt_dbscan_y_segments: n too large");
        return TDT_E_UNSUPPORTED;
    }
    const int n = (int)n_;
    int64_t ymin = y[0], ymax = y[0];
    for (int i = 0; i < n; i++) {
        ymin = std::min(ymin, y[i]);
        ymax = std::max(ymax, y[i]);
        if (seg[i] < 0 || seg[i] >= nseg) {
            tdt_set_error("tdt_dbscan_y_segments: segment %d of member %d is outside [0, %d)", seg[i], i, nseg);
            return TDT_E_ARG;
        }
    }
    if ((unsigned __int128)((__int128)ymax - ymin) > 0xfffffffeull) {
        tdt_set_error("tdt_dbscan_y_segments: coordinate span >= 2^32 is outside the device path's domain");
        return TDT_E_UNSUPPORTED;
    }
    // numpy: int64 distance < python number  <=>  d < ceil(eps) for integer d >= 0
    ull e = 0;
    if (eps > 0) e = eps >= 8589934592.0 ? (1ull << 33) : (ull)std::ceil(eps);
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t N = (size_t)n, G = (size_t)nseg;
    void *h = nullptr, *d = nullptr;
    int rc = tdt_pinned(ctx, 1, N * 16 + 64, &h);
    if (rc) return rc;
    const int ntile = (n + 1023) / 1024 + 1, gtile = (nseg + 1023) / 1024 + 1;
    const size_t total = 2 * al(N * 4) /*y, seg*/ + 2 * al(N * 8) /*keep, out*/ + 2 * al(N * 8) /*keys*/ + 2 * al(N * 4) /*vals*/ + 2 * al(N * 4) /*P,S*/ +
                         4 * al(G * 4 + 4) /*lo, hi, extras, extras_incl*/ + al((size_t)(ntile + gtile) * 4 + 64) + 256;
    rc = tdt_scratch(ctx, 5, total, &d);
    if (rc) return rc;
    char *p = (char *)d;
    auto carve = [&](size_t b) {
        void *r = p;
        p += al(b);
        return r;
    };
    unsigned *dy = (unsigned *)carve(N * 4);
    int *dseg = (int *)carve(N * 4);
    double *dkeep = (double *)carve(N * 8);
    double *dout = (double *)carve(N * 8);
    ull *dk0 = (ull *)carve(N * 8), *dk1 = (ull *)carve(N * 8);
    unsigned *dv0 = (unsigned *)carve(N * 4), *dv1 = (unsigned *)carve(N * 4);
    unsigned *dP = (unsigned *)carve(N * 4), *dS = (unsigned *)carve(N * 4);
    int *dlo = (int *)carve(G * 4 + 4), *dhi = (int *)carve(G * 4 + 4);
    unsigned *dex = (unsigned *)carve(G * 4 + 4), *dexi = (unsigned *)carve(G * 4 + 4);
    unsigned *dts = (unsigned *)carve((size_t)(ntile + gtile) * 4 + 64);
    long long *dlast = (long long *)carve(64);
    TDT_HIP(hipStreamSynchronize(st));                      // an earlier call may still be reading the pinned block
    unsigned *hy = (unsigned *)h;
    int *hs = (int *)(hy + N);
    double *hk = (double *)((char *)h + N * 8);
    for (int i = 0; i < n; i++) {
        hy[i] = (unsigned)(y[i] - ymin);
        hs[i] = seg[i];
        hk[i] = keep[i];
    }
    TDT_HIP(hipMemcpyAsync(dy, hy, N * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dseg, hs, N * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dkeep, hk, N * 8, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemsetAsync(dlo, 0, al(G * 4 + 4) * 2, st));                  // segments without members: lo = hi = 0
    const int blocks = (n + YS_THREADS - 1) / YS_THREADS;
    hipLaunchKernelGGL(ys_keys, dim3(blocks), dim3(YS_THREADS), 0, st, (const unsigned *)dy, (const int *)dseg, n, dk0, dv0);
    TDT_CHECK_LAUNCH();
    ull mask = ymax > ymin ? ((1ull << tdt_ceil_log2_u64((uint64_t)(ymax - ymin) + 1)) - 1ull) : 0ull;
    if (nseg > 1) mask |= ((1ull << tdt_ceil_log2_u64((uint64_t)nseg)) - 1ull) << 32;
    ull *ks = nullptr;
    unsigned *vs = nullptr;
    rc = tdt_radix_sort_pairs(ctx, dk0, dv0, dk1, dv1, N, mask, &ks, &vs);
    if (rc) return rc;
    hipLaunchKernelGGL(ys_flags, dim3(blocks), dim3(YS_THREADS), 0, st, (const ull *)ks, n, m, e, dP, dS, dlo, dhi);
    TDT_CHECK_LAUNCH();
    rc = tdt_scan_u32_inclusive(ctx, dP, n, dts);
    if (rc) return rc;
    rc = tdt_scan_u32_inclusive(ctx, dS, n, dts);
    if (rc) return rc;
    if (nseg) {
        hipLaunchKernelGGL(ys_extras, dim3((nseg + YS_THREADS - 1) / YS_THREADS), dim3(YS_THREADS), 0, st, (const unsigned *)dS, (const int *)dlo,
                           (const int *)dhi, nseg, dex);
        TDT_CHECK_LAUNCH();
        TDT_HIP(hipMemcpyAsync(dexi, dex, G * 4, hipMemcpyDeviceToDevice, st));
        rc = tdt_scan_u32_inclusive(ctx, dexi, nseg, dts + ntile);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ys_final, dim3(blocks), dim3(YS_THREADS), 0, st, (const ull *)ks, (const unsigned *)vs, n, m, (const unsigned *)dP,
                       (const unsigned *)dS, (const int *)dlo, (const unsigned *)dexi, (const unsigned *)dex, nseg, (const double *)dkeep,
                       (long long)cluster_id, dout, dlast);
    TDT_CHECK_LAUNCH();
    long long lid = cluster_id;
    TDT_HIP(hipMemcpyAsync(out, dout, N * 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(&lid, dlast, 8, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (last_id) *last_id = lid;
    return TDT_OK;
}
