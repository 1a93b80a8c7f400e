// Signal clustering ("DBSCAN") for gfx950 (MI355X).
//
// The reference's DBSCAN.py is two 1-D sliding-window passes driven by a sequential run-labelling
// state machine (x_coordinate_clustering DBSCAN.py:33-64, y_coordinate_clustering :66-123).  Its
// closed form is data parallel:
//   x pass   p[i]  = (i <= n-m) && max_{j in (i, min(i+m,n-1)]} |x_j - x_i| < eps
//            runs  = maximal stretches of consecutive true p, numbered 0,1,.. (an inclusive scan of run starts)
//            lab[k]= id of the run holding j* = the largest j <= k with p[j], if k - j* <= m-1, else -1
//   y pass   every x-cluster is a contiguous index range; its members are stably sorted by y, the
//            same run labelling is applied with window m-1, sub-run 1 keeps the x id and sub-run
//            s > 1 becomes (R-1) + #extra sub-runs of earlier x-clusters + (s-1).
// All of it is integer compares, prefix sums and a segmented sort: HBM/latency-bound, no MFMA.
// Several independent (chrA,chrB) buckets are processed by the same launches (ids restart per bucket).
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_dbscan;
const char *const tdt_variant_dbscan = ""
#ifdef DB_BUCKET_PERLANE
    " DB_BUCKET_PERLANE"
#endif
#ifdef DT_PERSIST
    " DT_PERSIST"
#endif
#ifdef DT_PROF
    " DT_PROF"
#endif
#ifdef DT_FTPB
    " DT_FTPB"
#endif
#ifdef DT_NW
    " DT_NW"
#endif
#ifdef DT_THREADS
    " DT_THREADS"
#endif
#ifdef DBF_THREADS
    " DBF_THREADS"
#endif
    ;

#include "tdt_common.h"
#include <chrono>

#include <atomic>
#include <thread>

#include <algorithm>
#include <cmath>

#define DB_THREADS 256
#define DB_ITEMS 4
#define DB_TILE (DB_THREADS * DB_ITEMS)
#define DB_SMALL 128  // x-clusters up to this many members are y-sorted by in-kernel rank counting

int tdt_radix_sort_pairs(tdt_ctx *ctx, unsigned long long *keys, unsigned *vals, unsigned long long *keys_tmp, unsigned *vals_tmp,
                         size_t n, unsigned long long bitmask, unsigned long long **out_keys, unsigned **out_vals);   // tdt_sort.hip

// largest b in [0, nb) with boff[b] <= i
__device__ __forceinline__ int db_bucket(const int *__restrict__ boff, int nb, int i);
// The bucket of position i when the wave's lanes hold positions inside [first, last] (first / last the same in every lane): the search
// is done ONCE, on the scalar unit, for `first`; a wave of 64-128 consecutive positions almost never contains a bucket boundary (300
// buckets in 10 M signals), and then one more scalar load settles it.  Lanes of a wave that does contain boundaries step forward from
// the first position's bucket.  (Per-lane binary searches — nine dependent vector loads per point — were what dbt_finish took 33 us for
// where the one-bucket dbt_finish1 takes 14.)
__device__ __forceinline__ int db_bucket_wave(const int *__restrict__ boff, int nb, int first, int last, int i) {
    if (nb == 1) return 0;
#ifdef DB_BUCKET_PERLANE   // measurement variant: every lane searches for itself, as before round 5
    return db_bucket(boff, nb, i);
#endif
    const int f = __builtin_amdgcn_readfirstlane(first), l = __builtin_amdgcn_readfirstlane(last);
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (boff[mid] <= f) lo = mid;
        else hi = mid;
    }
    int b = lo;
    if (lo + 1 < nb && boff[lo + 1] <= l)                  // (wave-uniform) a boundary inside the wave's range
        while (b + 1 < nb && boff[b + 1] <= i) b++;
    return b;
}
__device__ __forceinline__ int db_bucket(const int *__restrict__ boff, int nb, int i) {
    if (nb == 1) return 0;
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (boff[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ unsigned db_absdiff(unsigned a, unsigned b) { return a > b ? a - b : b - a; }

#include "tdt_dbscan_fused.h"
#include "tdt_dbscan_tile.h"

// workgroups of a dbt_tile launch: what the chip holds at once (eight 256-thread workgroups per CU), or one per tile if that is fewer
static inline int dbt_grid(const tdt_ctx *ctx, int ntiles) {
#ifndef DT_PERSIST
    (void)ctx;
    return ntiles;
#else
    const int resident = ctx->num_cu * 8;
    return ntiles < resident ? ntiles : resident;
#endif
}

// ---------------------------------------------------------------------------------------- scan
// In-place inclusive scan of a u32 array: reduce tiles -> scan the tile sums (one block) -> apply.
__global__ __launch_bounds__(DB_THREADS) void scan_reduce(const unsigned *__restrict__ v, int n, unsigned *__restrict__ tsum) {
    __shared__ unsigned red[DB_THREADS / 64];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * DB_TILE + tid * DB_ITEMS;
    unsigned s = 0;
    if (i0 + DB_ITEMS <= n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(v + i0);
        s = q.x + q.y + q.z + q.w;
    } else {
        for (int j = 0; j < DB_ITEMS; j++)
            if (i0 + j < n) s += v[i0 + j];
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) tsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of tsum[0..nt) in place, single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void scan_tiles(unsigned *tsum, int nt) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
        const int i = base + tid;
        const unsigned v = i < nt ? tsum[i] : 0;
        unsigned s = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(s, d);
            if (lane >= d) s += t;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const unsigned carry = carry_s;
        if (i < nt) tsum[i] = carry + woff + s - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + s;
        __syncthreads();
    }
}

__global__ __launch_bounds__(DB_THREADS) void scan_apply(unsigned *__restrict__ v, int n, const unsigned *__restrict__ tsum) {
    __shared__ unsigned wsum[DB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * DB_TILE + tid * DB_ITEMS;
    unsigned a[DB_ITEMS];
    const bool full = i0 + DB_ITEMS <= n;
    if (full) {
        const uint4 q = *reinterpret_cast<const uint4 *>(v + i0);
        a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w;
    } else {
        for (int j = 0; j < DB_ITEMS; j++) a[j] = i0 + j < n ? v[i0 + j] : 0;
    }
    a[1] += a[0]; a[2] += a[1]; a[3] += a[2];
    unsigned s = a[3];
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(s, d);
        if (lane >= d) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    unsigned off = tsum[blockIdx.x] + s - a[3];
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (full) {
        *reinterpret_cast<uint4 *>(v + i0) = make_uint4(a[0] + off, a[1] + off, a[2] + off, a[3] + off);
    } else {
        for (int j = 0; j < DB_ITEMS; j++)
            if (i0 + j < n) v[i0 + j] = a[j] + off;
    }
}

// -------------------------------------------------------------------------------------- x pass
// p[i] (DBSCAN.py:41-51) and run-start flags (the `cluster` boolean of :52-62)
__global__ __launch_bounds__(DB_THREADS) void dbx_flags(const unsigned *__restrict__ x, int n, const int *__restrict__ boff,
                                                        int nb, unsigned long long eps, int m,
                                                        unsigned char *__restrict__ px, unsigned *__restrict__ sx) {
    const int i0 = (blockIdx.x * DB_THREADS + threadIdx.x) * DB_ITEMS;
    if (i0 >= n) return;
    int b = db_bucket(boff, nb, i0 > 0 ? i0 - 1 : 0);
    bool prev = false;
    for (int i = (i0 > 0 ? i0 - 1 : 0); i < i0 + DB_ITEMS && i < n; i++) {
        while (i >= boff[b + 1]) b++;
        const int bend = boff[b + 1];
        bool p = false;
        if (i + m <= bend) {  // the loop `for i in range(0, len(data)-m+1)` (:39)
            const int hi = min(i + m, bend - 1);  // data[i+1:i+m+1] truncates at the array end (:43)
            const unsigned xi = x[i];
            unsigned maxd = 0;
            for (int j = i + 1; j <= hi; j++) maxd = max(maxd, db_absdiff(x[j], xi));
            p = (unsigned long long)maxd < eps;
        }
        if (i >= i0) {
            px[i] = p;
            sx[i] = (p && !prev) ? 1u : 0u;
        }
        prev = p;
    }
}

// lab[k] = (scan of run starts at j*) - 1; also the number of runs before every bucket
__global__ __launch_bounds__(DB_THREADS) void dbx_labels(const unsigned char *__restrict__ px, const unsigned *__restrict__ sx_incl,
                                                         int n, int m, const int *__restrict__ boff, int nb,
                                                         int *__restrict__ xlab, unsigned *__restrict__ runbase) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k >= n) return;
    int lab = -1;
    for (int j = k; j >= 0 && j > k - m; j--) {
        if (px[j]) {
            lab = (int)sx_incl[j] - 1;
            break;
        }
    }
    xlab[k] = lab;
    const int b = db_bucket(boff, nb, k);
    if (k + 1 == boff[b + 1]) {
        const unsigned s = sx_incl[k];
        for (int bb = b + 1; bb <= nb && boff[bb] == k + 1; bb++) runbase[bb] = s;
    }
}

__global__ __launch_bounds__(DB_THREADS) void db_segments(const int *__restrict__ xlab, int n, int *__restrict__ seg_start,
                                                          int *__restrict__ seg_end) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k >= n) return;
    const int l = xlab[k];
    if (l < 0) return;
    if (k == 0 || xlab[k - 1] != l) seg_start[l] = k;
    if (k == n - 1 || xlab[k + 1] != l) seg_end[l] = k + 1;
}

// x-only result (x_coordinate_clustering's return value)
__global__ __launch_bounds__(DB_THREADS) void dbx_final(const int *__restrict__ xlab, int n, const int *__restrict__ boff, int nb,
                                                        const unsigned *__restrict__ runbase, double *__restrict__ labels,
                                                        long long *__restrict__ last_id) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k < nb && last_id) last_id[k] = (long long)(runbase[k + 1] - runbase[k]) - 1;
    if (k >= n) return;
    const int l = xlab[k];
    labels[k] = l < 0 ? -1.0 : (double)(l - (int)runbase[db_bucket(boff, nb, k)]);
}

// One word back to the host without a stream synchronisation: the value goes to pinned host memory, then a sequence number;
// the host spins on the sequence number (a hipStreamSynchronize wake-up costs ~40 us, a third of a 5 M-point pass).
__global__ void db_signal_host(const unsigned *__restrict__ word, volatile unsigned *host, unsigned seq) {
    host[0] = *word;
    __threadfence_system();
    host[1] = seq;
}

// -------------------------------------------------------------------------------------- y pass
// stable sort by y inside every x-cluster (DBSCAN.py:76-81): rank counting for small clusters; members of
// larger clusters are flagged and go through the radix sort below
__global__ __launch_bounds__(DB_THREADS) void dby_rank(const int *__restrict__ xlab, const unsigned *__restrict__ y, int n,
                                                       const int *__restrict__ seg_start, const int *__restrict__ seg_end,
                                                       unsigned *__restrict__ ys, unsigned *__restrict__ ord,
                                                       unsigned *__restrict__ lflag, unsigned *__restrict__ anylarge) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k >= n) return;
    const int l = xlab[k];
    unsigned large = 0;
    if (l < 0) {
        ys[k] = 0;
        ord[k] = k;
    } else {
        const int s0 = seg_start[l], s1 = seg_end[l];
        const unsigned yk = y[k];
        if (s1 - s0 <= DB_SMALL) {
            int rank = 0;
            // rank = members sorting before k.  The wave runs as long as its largest cluster, so the cost is loop
            // overhead x trip count: 4 members per trip (clamped loads, no per-member branch)
            for (int j = s0; j < s1; j += 4) {
                const unsigned v0 = y[j], v1 = y[min(j + 1, s1 - 1)], v2 = y[min(j + 2, s1 - 1)], v3 = y[min(j + 3, s1 - 1)];
                rank += (v0 < yk) || (v0 == yk && j < k);
                rank += (j + 1 < s1) && ((v1 < yk) || (v1 == yk && j + 1 < k));
                rank += (j + 2 < s1) && ((v2 < yk) || (v2 == yk && j + 2 < k));
                rank += (j + 3 < s1) && ((v3 < yk) || (v3 == yk && j + 3 < k));
            }
            ys[s0 + rank] = yk;
            ord[s0 + rank] = k;
        } else {
            large = 1;
            ys[k] = yk;   // placeholders: the speculative y pass runs before the large clusters are sorted and must
            ord[k] = k;   // scatter in bounds; dby_large_scatter overwrites both
            if (k == s0) *anylarge = 1u;
        }
    }
    lflag[k] = large;
}

// members of large clusters, compacted in position order: key = cluster id << 32 | y, value = position
__global__ __launch_bounds__(DB_THREADS) void dby_large_compact(const int *__restrict__ xlab, const unsigned *__restrict__ y, int n,
                                                                const unsigned *__restrict__ lincl, unsigned long long *__restrict__ ck,
                                                                unsigned *__restrict__ cv, unsigned *__restrict__ cpos) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k >= n) return;
    const unsigned inc = lincl[k], prev = k ? lincl[k - 1] : 0u;
    if (inc != prev) {
        ck[prev] = ((unsigned long long)(unsigned)xlab[k] << 32) | y[k];
        cv[prev] = (unsigned)k;
        cpos[prev] = (unsigned)k;
    }
}

// clusters are contiguous and the compaction kept position order, so the j-th sorted pair belongs at the j-th position
__global__ __launch_bounds__(DB_THREADS) void dby_large_scatter(const unsigned long long *__restrict__ ksorted, const unsigned *__restrict__ vsorted,
                                                                const unsigned *__restrict__ cpos, int nl, unsigned *__restrict__ ys,
                                                                unsigned *__restrict__ ord) {
    const int j = blockIdx.x * DB_THREADS + threadIdx.x;
    if (j >= nl) return;
    const unsigned p = cpos[j];
    ys[p] = (unsigned)ksorted[j];
    ord[p] = vsorted[j];
}

// window test on the sorted y of each x-cluster (DBSCAN.py:90-99) and sub-run starts (:101-110)
__global__ __launch_bounds__(DB_THREADS) void dby_flags(const int *__restrict__ xlab, int n, const int *__restrict__ seg_start,
                                                        const int *__restrict__ seg_end, const unsigned *__restrict__ ys,
                                                        unsigned long long eps, int m, unsigned char *__restrict__ py,
                                                        unsigned *__restrict__ sy) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const int l = xlab[i];
    bool p = false, prev = false;
    if (l >= 0) {
        const int s0 = seg_start[l], s1 = seg_end[l];
        if (i + m <= s1) p = (unsigned long long)(ys[i + m - 1] - ys[i]) < eps;  // next = y[i+1:i+m], sorted => max is the last
        if (i > s0 && (i - 1) + m <= s1) prev = (unsigned long long)(ys[i + m - 2] - ys[i - 1]) < eps;
    }
    py[i] = p;
    sy[i] = (p && !prev) ? 1u : 0u;
}

// extra sub-runs of every x-cluster, stored at the cluster's first position (`cluster_id += sub_cluster_id-1`, :121-122)
__global__ __launch_bounds__(DB_THREADS) void dby_extras(const int *__restrict__ xlab, int n, const int *__restrict__ seg_start,
                                                         const int *__restrict__ seg_end, const unsigned *__restrict__ sy_incl,
                                                         unsigned *__restrict__ ex) {
    const int k = blockIdx.x * DB_THREADS + threadIdx.x;
    if (k >= n) return;
    const int l = xlab[k];
    unsigned e = 0;
    if (l >= 0 && seg_start[l] == k) {
        const unsigned sr = sy_incl[seg_end[l] - 1] - (k > 0 ? sy_incl[k - 1] : 0u);
        e = sr > 1 ? sr - 1 : 0;
    }
    ex[k] = e;
}

// relabel (DBSCAN.py:112-119) and scatter back to the input order
__global__ __launch_bounds__(DB_THREADS) void dby_final(const int *__restrict__ xlab, int n, int m, const int *__restrict__ seg_start,
                                                        const unsigned char *__restrict__ py, const unsigned *__restrict__ sy_incl,
                                                        const unsigned *__restrict__ ex_incl, const unsigned *__restrict__ ord,
                                                        const int *__restrict__ boff, int nb, const unsigned *__restrict__ runbase,
                                                        double *__restrict__ labels, long long *__restrict__ last_id) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i < nb && last_id) {
        const int b0 = boff[i], b1 = boff[i + 1];
        const long long extras = (long long)(b1 > 0 ? ex_incl[b1 - 1] : 0u) - (long long)(b0 > 0 ? ex_incl[b0 - 1] : 0u);
        last_id[i] = (long long)(runbase[i + 1] - runbase[i]) - 1 + extras;
    }
    if (i >= n) return;
    const int l = xlab[i];
    if (l < 0) {
        labels[i] = -1.0;  // ord[i] == i for noise
        return;
    }
    double lab = -1.0;
    for (int j = i; j >= 0 && j > i - m; j--) {
        if (py[j]) {
            const int s0 = seg_start[l];
            const unsigned s = sy_incl[j] - (s0 > 0 ? sy_incl[s0 - 1] : 0u);
            const int b = db_bucket(boff, nb, i);
            const unsigned rb = runbase[b];
            if (s == 1) {
                lab = (double)((unsigned)l - rb);
            } else {
                const int b0 = boff[b];
                const unsigned xoff = (s0 > 0 ? ex_incl[s0 - 1] : 0u) - (b0 > 0 ? ex_incl[b0 - 1] : 0u);
                lab = (double)((long long)(runbase[b + 1] - rb) - 1 + (long long)xoff + (long long)(s - 1));
            }
            break;
        }
    }
    labels[ord[i]] = lab;
}

// ------------------------------------------------------------------------------------------ host
static inline size_t db_align(size_t v) { return (v + 255) & ~(size_t)255; }

static int db_scan_inplace(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum) {
    const int nt = (n + DB_TILE - 1) / DB_TILE;
    hipLaunchKernelGGL(scan_reduce, dim3(nt), dim3(DB_THREADS), 0, ctx->stream, (const unsigned *)d_v, n, d_tsum);
    hipLaunchKernelGGL(scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, d_tsum, nt);
    hipLaunchKernelGGL(scan_apply, dim3(nt), dim3(DB_THREADS), 0, ctx->stream, d_v, n, (const unsigned *)d_tsum);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

int tdt_scan_u32_inclusive(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum) { return db_scan_inplace(ctx, d_v, n, d_tsum); }

// y-sort of the clusters larger than DB_SMALL (runs only when dby_rank flagged any)
static int db_sort_large(tdt_ctx *ctx, const int *d_xlab, const unsigned *d_y, int n, unsigned *d_lflag, unsigned *d_tsum,
                         unsigned long long *d_k0, unsigned long long *d_k1, unsigned *d_v0, unsigned *d_v1, unsigned *d_cpos,
                         unsigned *d_ys, unsigned *d_ord) {
    hipStream_t st = ctx->stream;
    int rc = db_scan_inplace(ctx, d_lflag, n, d_tsum);
    if (rc) return rc;
    unsigned nl = 0;
    TDT_HIP(hipMemcpyAsync(&nl, d_lflag + (n - 1), 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (!nl) return TDT_OK;
    const int blocks1 = (n + DB_THREADS - 1) / DB_THREADS;
    hipLaunchKernelGGL(dby_large_compact, dim3(blocks1), dim3(DB_THREADS), 0, st, d_xlab, d_y, n, (const unsigned *)d_lflag, d_k0, d_v0, d_cpos);
    TDT_CHECK_LAUNCH();
    unsigned long long mask = 0xffffffffull;                                 // y: all 32 bits
    mask |= ((1ull << tdt_ceil_log2_u64((uint64_t)n + 1)) - 1ull) << 32;     // cluster id < number of runs <= n
    unsigned long long *ks = nullptr;
    unsigned *vs = nullptr;
    rc = tdt_radix_sort_pairs(ctx, d_k0, d_v0, d_k1, d_v1, nl, mask, &ks, &vs);
    if (rc) return rc;
    hipLaunchKernelGGL(dby_large_scatter, dim3((nl + DB_THREADS - 1) / DB_THREADS), dim3(DB_THREADS), 0, st, (const unsigned long long *)ks,
                       (const unsigned *)vs, (const unsigned *)d_cpos, (int)nl, d_ys, d_ord);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_dbscan_device(tdt_ctx *ctx, const uint32_t *d_x, const uint32_t *d_y, size_t n_, const int64_t *bucket_off,
                                 int nb, uint64_t eps, int m, int mode, double *d_labels, int64_t *d_last_id) {
    if (!ctx || nb < 1 || !bucket_off || m < 2 || (mode != 0 && mode != 1)) {
        tdt_set_error("tdt_dbscan_device: bad argument (m must be >= 2: the reference's max() of an empty window raises)");
        return TDT_E_ARG;
    }
    if (n_ >= 0x7fffffffull) {
        tdt_set_error("tdt_dbscan_device: n too large");
        return TDT_E_UNSUPPORTED;
    }
    const int n = (int)n_;
    if (bucket_off[0] != 0 || bucket_off[nb] != (int64_t)n) {
        tdt_set_error("tdt_dbscan_device: bucket_off must start at 0 and end at n");
        return TDT_E_ARG;
    }
    for (int b = 0; b < nb; b++)
        if (bucket_off[b + 1] < bucket_off[b]) {
            tdt_set_error("tdt_dbscan_device: bucket_off must be non-decreasing");
            return TDT_E_ARG;
        }
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // small control block first (needed even for n == 0)
    const size_t sz_boff = db_align((size_t)(nb + 1) * 4);
    const size_t sz_rb = db_align((size_t)(nb + 1) * 4);
    const int nt = n ? (n + DB_TILE - 1) / DB_TILE : 1;
    const size_t N = (size_t)(n ? n : 1);
    const size_t nlarge_cap = N / DB_SMALL + 1;
    size_t total = sz_boff + sz_rb + db_align(256) /*counters*/ + db_align((size_t)nt * 4) +
                   2 * db_align(N) /*px,py*/ + 3 * db_align(N * 4) /*sx,sy,ex*/ + 3 * db_align(N * 4) /*xlab,seg_start,seg_end*/ +
                   2 * db_align(N * 4) /*ys,ord*/ + 2 * db_align(N * 8) /*sort keys in/out*/ + 4 * db_align(N * 4) /*sort vals, cpos, lflag*/ +
                   0 * nlarge_cap;
    void *base = nullptr;
    int rc = tdt_scratch(ctx, 3, total, &base);
    if (rc) return rc;
    char *p = (char *)base;
    auto carve = [&](size_t bytes) {
        void *r = p;
        p += db_align(bytes);
        return r;
    };
    int *d_boff = (int *)carve((size_t)(nb + 1) * 4);
    unsigned *d_runbase = (unsigned *)carve((size_t)(nb + 1) * 4);
    unsigned *d_cnt = (unsigned *)carve(256);
    unsigned *d_tsum = (unsigned *)carve((size_t)nt * 4);
    unsigned char *d_px = (unsigned char *)carve(N);
    unsigned char *d_py = (unsigned char *)carve(N);
    unsigned *d_sx = (unsigned *)carve(N * 4);
    unsigned *d_sy = (unsigned *)carve(N * 4);
    unsigned *d_ex = (unsigned *)carve(N * 4);
    int *d_xlab = (int *)carve(N * 4);
    int *d_seg0 = (int *)carve(N * 4);
    int *d_seg1 = (int *)carve(N * 4);
    unsigned *d_ys = (unsigned *)carve(N * 4);
    unsigned *d_ord = (unsigned *)carve(N * 4);
    unsigned long long *d_key = (unsigned long long *)carve(N * 8);
    unsigned long long *d_ksorted = (unsigned long long *)carve(N * 8);
    unsigned *d_v0 = (unsigned *)carve(N * 4);
    unsigned *d_v1 = (unsigned *)carve(N * 4);
    unsigned *d_cpos = (unsigned *)carve(N * 4);
    unsigned *d_lflag = (unsigned *)carve(N * 4);

    // bucket offsets: stage through pinned memory so the copy is truly asynchronous.  (Not needed by the one-bucket tile path,
    // which therefore never waits for the stream.)
    bool prologue_done = false;
    auto prologue = [&]() -> int {
        if (prologue_done) return TDT_OK;
        prologue_done = true;
        void *h_stage = nullptr;
        int rc2 = tdt_pinned(ctx, 0, (size_t)(nb + 1) * 4 + 64, &h_stage);
        if (rc2) return rc2;
        TDT_HIP(hipStreamSynchronize(st));  // previous call may still be reading the pinned block
        int *h_boff = (int *)h_stage;
        for (int b = 0; b <= nb; b++) h_boff[b] = (int)bucket_off[b];
        TDT_HIP(hipMemcpyAsync(d_boff, h_boff, (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st));
        TDT_HIP(hipMemsetAsync(d_runbase, 0, (size_t)(nb + 1) * 4, st));
        TDT_HIP(hipMemsetAsync(d_cnt, 0, 256, st));
        return TDT_OK;
    };

    const int blocks1 = n ? (n + DB_THREADS - 1) / DB_THREADS : 1;
    const int blocks4 = n ? (n + DB_TILE - 1) / DB_TILE : 1;
    const int blocks_nb = (std::max(n, nb) + DB_THREADS - 1) / DB_THREADS;
    if (n == 0) {
        // empty input: every bucket reports cluster_id -1
        rc = prologue();
        if (rc) return rc;
        hipLaunchKernelGGL(dbx_final, dim3(blocks_nb), dim3(DB_THREADS), 0, st, (const int *)d_xlab, 0, (const int *)d_boff, nb,
                           (const unsigned *)d_runbase, d_labels, (long long *)d_last_id);
        TDT_CHECK_LAUNCH();
        return TDT_OK;
    }
    const bool fused = m <= DBF_M_MAX;
    static const bool no_tile = getenv("TIDDIT_DBSCAN_LAUNCHES") != nullptr;     // measurement switch: the multi-launch path only
    if (fused && !no_tile && n < 0x7fff0000) {
        // tile-resident pass (tdt_dbscan_tile.h): three launches; falls through to the multi-launch path below when an
        // x-cluster is too large for it
        const int ntt = (n + DT_T - 1) / DT_T;
        const size_t sz_flags = 256, sz_agg = db_align((size_t)ntt * 4), sz_b = db_align((size_t)(nb + 1) * 4);
        void *tb = nullptr;
        const size_t sz_grp = db_align((size_t)2 * DT_GRPMAX * 4);
        // the status block and the two group-sum arrays keep their contents from call to call: their own fixed-size allocation
        void *ts = nullptr;
        rc = tdt_scratch(ctx, 20, sz_flags + 2 * sz_grp, &ts);
        if (rc) return rc;
        unsigned *t_flags = (unsigned *)ts;
        unsigned *t_grp0 = (unsigned *)((char *)ts + sz_flags);
        unsigned *t_grp1 = (unsigned *)((char *)ts + sz_flags + sz_grp);
        const size_t sz_code = db_align((size_t)n * 2 + 16);
        rc = tdt_scratch(ctx, 21, 4 * sz_b + 2 * sz_agg + sz_code, &tb);
        if (rc) return rc;
        char *q = (char *)tb;
        unsigned short *t_code = (unsigned short *)q; q += sz_code;
        unsigned *t_brun = (unsigned *)q; q += sz_b;
        unsigned *t_bext = (unsigned *)q; q += sz_b;
        unsigned *t_aggR = (unsigned *)q; q += sz_agg;
        unsigned *t_aggE = (unsigned *)q; q += sz_agg;
        unsigned *t_runbase = (unsigned *)q; q += sz_b;
        unsigned *t_extbase = (unsigned *)q; q += sz_b;
        if (nb > 1) {
            rc = prologue();
            if (rc) return rc;
        }
        if (ctx->tile_flags_zeroed != ts) {       // first use of this block; afterwards the kernel that reports the status re-zeroes it
            TDT_HIP(hipMemsetAsync(t_flags, 0, sz_flags, st));
            TDT_HIP(hipMemsetAsync(t_grp0, 0, 2 * sz_grp, st));
            ctx->tile_flags_zeroed = ts;
        }
        DtParams TP;
        TP.x = d_x;
        TP.y = d_y;
        TP.n = n;
        TP.boff = d_boff;
        TP.nb = nb;
        TP.eps32 = eps > 0xffffffffull ? 0xffffffffu : (unsigned)eps;
        TP.wide = eps > 0xffffffffull;
        TP.m = m;
        TP.code = t_code;
        TP.aggR = t_aggR;
        TP.aggE = t_aggE;
        TP.brun = t_brun;
        TP.bext = t_bext;
        TP.flags = t_flags;
        const bool odd = nb == 1 && (ctx->tile_calls++ & 1u) != 0;     // (only the one-bucket kernels touch the group sums)
        TP.grp = odd ? t_grp1 : t_grp0;
        void *hp = nullptr;
        rc = tdt_pinned(ctx, 2, 64, &hp);
        if (rc) return rc;
        volatile unsigned *hw = (volatile unsigned *)hp;
        static std::atomic<unsigned> tile_seq{0};
        unsigned seq = ++tile_seq;
        if (seq == 0) seq = ++tile_seq;                                    // never 0
        hw[1] = 0;
        const int tgrid = dbt_grid(ctx, ntt);
        if (nb == 1 && mode == 0) hipLaunchKernelGGL((dbt_tile<true, false>), dim3(tgrid), dim3(DT_THREADS), 0, st, TP);
        else if (nb == 1) hipLaunchKernelGGL((dbt_tile<true, true>), dim3(tgrid), dim3(DT_THREADS), 0, st, TP);
        else if (mode == 0) hipLaunchKernelGGL((dbt_tile<false, false>), dim3(tgrid), dim3(DT_THREADS), 0, st, TP);
        else hipLaunchKernelGGL((dbt_tile<false, true>), dim3(tgrid), dim3(DT_THREADS), 0, st, TP);
        TDT_CHECK_LAUNCH();
        if (nb == 1) {
            ctx->tile_groups_max = std::max(ctx->tile_groups_max, (ntt + DT_GRP - 1) / DT_GRP);
            hipLaunchKernelGGL(dbt_finish1, dim3((ntt + DT_FTPB - 1) / DT_FTPB), dim3(256), 0, st, (const unsigned short *)t_code, d_labels, n, (const int *)nullptr,
                               (const unsigned *)t_aggR, (const unsigned *)t_aggE, ntt, (const unsigned *)TP.grp, odd ? t_grp0 : t_grp1, ctx->tile_groups_max,
                               (long long *)d_last_id, 0ll, 0, t_flags, hw, seq);
        } else {
            hipLaunchKernelGGL(dbt_scan, dim3(1), dim3(1024), 0, st, t_aggR, t_aggE, ntt, (const int *)d_boff, nb, n, (const unsigned *)t_brun,
                               (const unsigned *)t_bext, t_runbase, t_extbase, (long long *)d_last_id, mode, t_flags, hw, seq);
            hipLaunchKernelGGL(dbt_finish, dim3((n + 1023) / 1024), dim3(256), 0, st, (const unsigned short *)t_code, d_labels, n, (const unsigned *)t_aggR,
                               (const unsigned *)t_aggE, (const int *)d_boff, nb, (const unsigned *)t_runbase, (const unsigned *)t_extbase);
        }
        TDT_CHECK_LAUNCH();
        // the one word that says whether the pass stands comes back through pinned memory (a hipStreamSynchronize wake-up costs
        // more than the whole pass); dbt_scan stores it as soon as the tile kernel is done
        bool seen = false;
        for (long spin = 0; spin < 4000000; spin++) {
            if (hw[1] == seq) {
                seen = true;
                break;
            }
            __builtin_ia32_pause();
        }
        if (!seen) TDT_HIP(hipStreamSynchronize(st));
        if (hw[0] == 0) return TDT_OK;
    }
    rc = prologue();
    if (rc) return rc;
    if (fused) {
        // ballot-mask tiles; cross-tile prefixes from a one-workgroup scan between launches
        const int ntf = (n + DBF_TILE - 1) / DBF_TILE;
        const size_t nw = (size_t)ntf * DBF_WORDS + 2;   // whole tiles: kernels store all 64 words of their tile
        const size_t mask_bytes = db_align(nw * 8);
        const size_t ctl_bytes = db_align(sizeof(DbfCtl)) + 3 * db_align((size_t)ntf * 8) + 7 * mask_bytes;
        void *cb = nullptr;
        rc = tdt_scratch(ctx, 8, ctl_bytes, &cb);
        if (rc) return rc;
        char *q = (char *)cb;
        DbfCtl *ctl = (DbfCtl *)q; q += db_align(sizeof(DbfCtl));
        ull *agg_x = (ull *)q; q += db_align((size_t)ntf * 8);
        ull *agg_1 = (ull *)q; q += db_align((size_t)ntf * 8);
        ull *agg_2 = (ull *)q; q += db_align((size_t)ntf * 8);
        ull *PM = (ull *)q; q += mask_bytes;
        ull *PY = (ull *)q; q += mask_bytes;
        ull *HM = (ull *)q; q += mask_bytes;
        ull *BM = (ull *)q; q += mask_bytes;
        ull *EM = (ull *)q; q += mask_bytes;
        ull *S1M = (ull *)q; q += mask_bytes;
        ull *FM = (ull *)q; q += mask_bytes;
        // (the control word and the three guard words of the mask arrays are zeroed by tile 0 of dbm_x_masks)
        if (nb == 1 && m <= 4)
            hipLaunchKernelGGL(dbm_x_masks<true>, dim3(ntf), dim3(DBF_THREADS), 0, st, d_x, n, (const int *)d_boff, nb, (ull)eps, m, PM, agg_x, PY, ctl);
        else
            hipLaunchKernelGGL(dbm_x_masks<false>, dim3(ntf), dim3(DBF_THREADS), 0, st, d_x, n, (const int *)d_boff, nb, (ull)eps, m, PM, agg_x, PY, ctl);
        if (ntf > DBM_INLINE_PREFIX_MAX) hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, st, agg_x, ntf);
        hipLaunchKernelGGL(dbm_x_labels, dim3(ntf), dim3(DBF_THREADS), 0, st, (const ull *)PM, (const ull *)agg_x, n, (const int *)d_boff, nb,
                           m, d_xlab, d_runbase, d_seg0, d_seg1);
        TDT_CHECK_LAUNCH();
        if (mode == 1) {
            hipLaunchKernelGGL(dbx_final, dim3(blocks_nb), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, (const int *)d_boff, nb,
                               (const unsigned *)d_runbase, d_labels, (long long *)d_last_id);
            TDT_CHECK_LAUNCH();
            return TDT_OK;
        }
        hipLaunchKernelGGL(dby_rank, dim3(blocks1), dim3(DB_THREADS), 0, st, (const int *)d_xlab, d_y, n, (const int *)d_seg0,
                           (const int *)d_seg1, d_ys, d_ord, d_lflag, &ctl->nlarge);
        TDT_CHECK_LAUNCH();
        // The y pass is enqueued right away on the assumption that no x-cluster exceeded DB_SMALL members (the
        // usual case), so the GPU never idles on a mid-pipeline readback; the counter is checked afterwards
        // and only then are the large clusters sorted and the y pass repeated.
        for (int attempt = 0; attempt < 2; attempt++) {
            hipLaunchKernelGGL(dbm_y_masks, dim3(ntf), dim3(DBF_THREADS), 0, st, (const int *)d_xlab, (const unsigned *)d_ys, n,
                               (const int *)d_boff, nb, (ull)eps, m, PY, HM, BM, agg_1);
            if (ntf > DBM_INLINE_PREFIX_MAX) hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, st, agg_1, ntf);
            hipLaunchKernelGGL(dbm_y_mid, dim3(ntf), dim3(DBF_THREADS), 0, st, (const ull *)PY, (const ull *)HM, (const ull *)BM,
                               (const ull *)agg_1, n, m, EM, S1M, FM, agg_2);
            if (ntf > DBM_INLINE_PREFIX_MAX) hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, st, agg_2, ntf);
            hipLaunchKernelGGL(dbm_y_final, dim3(ntf), dim3(DBF_THREADS), 0, st, (const int *)d_xlab, (const unsigned *)d_ord, (const ull *)BM,
                               (const ull *)EM, (const ull *)S1M, (const ull *)FM, (const ull *)agg_2, n, (const int *)d_boff, nb,
                               (const unsigned *)d_runbase, d_labels, (long long *)d_last_id);
            TDT_CHECK_LAUNCH();
            if (attempt == 1) break;
            unsigned nlarge = 0;
            {
                void *hp = nullptr;
                rc = tdt_pinned(ctx, 2, 64, &hp);
                if (rc) return rc;
                volatile unsigned *hw = (volatile unsigned *)hp;
                static std::atomic<unsigned> seq_counter{0};
                unsigned seq = ++seq_counter;
                if (seq == 0) seq = ++seq_counter;                                    // never 0
                hw[1] = 0;
                hipLaunchKernelGGL(db_signal_host, dim3(1), dim3(1), 0, st, (const unsigned *)&ctl->nlarge, hw, seq);
                TDT_CHECK_LAUNCH();
                bool seen = false;
                for (long spin = 0; spin < 4000000; spin++) {     // a few milliseconds at most, then the ordinary wait
                    if (hw[1] == seq) {
                        seen = true;
                        break;
                    }
                    __builtin_ia32_pause();
                }
                if (!seen) TDT_HIP(hipStreamSynchronize(st));
                nlarge = hw[0];
            }
            if (!nlarge) break;
            rc = db_sort_large(ctx, d_xlab, d_y, n, d_lflag, d_tsum, d_key, d_ksorted, d_v0, d_v1, d_cpos, d_ys, d_ord);
            if (rc) return rc;
        }
        if (d_last_id && nb > 1)
            hipLaunchKernelGGL(dbf_empty_buckets, dim3((nb + 255) / 256), dim3(256), 0, st, (const int *)d_boff, nb, (long long *)d_last_id);
        TDT_CHECK_LAUNCH();
        return TDT_OK;
    }
    hipLaunchKernelGGL(dbx_flags, dim3(blocks4), dim3(DB_THREADS), 0, st, d_x, n, (const int *)d_boff, nb,
                       (unsigned long long)eps, m, d_px, d_sx);
    TDT_CHECK_LAUNCH();
    rc = db_scan_inplace(ctx, d_sx, n, d_tsum);
    if (rc) return rc;
    hipLaunchKernelGGL(dbx_labels, dim3(blocks1), dim3(DB_THREADS), 0, st, (const unsigned char *)d_px, (const unsigned *)d_sx, n, m,
                       (const int *)d_boff, nb, d_xlab, d_runbase);
    TDT_CHECK_LAUNCH();
    if (mode == 1) {
        hipLaunchKernelGGL(dbx_final, dim3(blocks_nb), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, (const int *)d_boff, nb,
                           (const unsigned *)d_runbase, d_labels, (long long *)d_last_id);
        TDT_CHECK_LAUNCH();
        return TDT_OK;
    }
    hipLaunchKernelGGL(db_segments, dim3(blocks1), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, d_seg0, d_seg1);
    hipLaunchKernelGGL(dby_rank, dim3(blocks1), dim3(DB_THREADS), 0, st, (const int *)d_xlab, d_y, n, (const int *)d_seg0,
                       (const int *)d_seg1, d_ys, d_ord, d_lflag, d_cnt);
    TDT_CHECK_LAUNCH();
    // x-clusters larger than DB_SMALL: one 4-byte readback decides whether the radix sort runs
    unsigned nlarge = 0;
    TDT_HIP(hipMemcpyAsync(&nlarge, d_cnt, 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (nlarge) {
        rc = db_sort_large(ctx, d_xlab, d_y, n, d_lflag, d_tsum, d_key, d_ksorted, d_v0, d_v1, d_cpos, d_ys, d_ord);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(dby_flags, dim3(blocks1), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, (const int *)d_seg0,
                       (const int *)d_seg1, (const unsigned *)d_ys, (unsigned long long)eps, m, d_py, d_sy);
    TDT_CHECK_LAUNCH();
    rc = db_scan_inplace(ctx, d_sy, n, d_tsum);
    if (rc) return rc;
    hipLaunchKernelGGL(dby_extras, dim3(blocks1), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, (const int *)d_seg0,
                       (const int *)d_seg1, (const unsigned *)d_sy, d_ex);
    TDT_CHECK_LAUNCH();
    rc = db_scan_inplace(ctx, d_ex, n, d_tsum);
    if (rc) return rc;
    hipLaunchKernelGGL(dby_final, dim3(blocks_nb), dim3(DB_THREADS), 0, st, (const int *)d_xlab, n, m, (const int *)d_seg0,
                       (const unsigned char *)d_py, (const unsigned *)d_sy, (const unsigned *)d_ex, (const unsigned *)d_ord,
                       (const int *)d_boff, nb, (const unsigned *)d_runbase, d_labels, (long long *)d_last_id);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

static uint64_t db_eps_u64(double eps) {
    // numpy: int64 distance < python number  <=>  d < ceil(eps) for integer d >= 0
    if (!(eps > 0)) return 0;  // also NaN: nothing is ever < NaN
    if (eps >= 8589934592.0) return 1ull << 33;
    return (uint64_t)ceil(eps);
}

extern "C" int tdt_dbscan(tdt_ctx *ctx, const int64_t *data, size_t n, size_t stride, double eps, int m, int mode,
                          double *labels, int64_t *last_id) {
    if (!ctx || (n && (!data || !labels)) || stride < 1 || (mode == 0 && stride < 2)) {
        tdt_set_error("tdt_dbscan: bad argument");
        return TDT_E_ARG;
    }
    if (m < 2) {
        tdt_set_error("tdt_dbscan: m must be >= 2 (the reference raises ValueError: max() arg is an empty sequence)");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    if (n == 0) {
        if (last_id) *last_id = -1;
        return TDT_OK;
    }
    // device coordinates are uint32 offsets from the column minimum
    int64_t xmin = data[0], xmax = data[0], ymin = 0, ymax = 0;
    if (stride >= 2) ymin = ymax = data[1];
    for (size_t i = 0; i < n; i++) {
        const int64_t xv = data[i * stride];
        xmin = xv < xmin ? xv : xmin;
        xmax = xv > xmax ? xv : xmax;
        if (stride >= 2) {
            const int64_t yv = data[i * stride + 1];
            ymin = yv < ymin ? yv : ymin;
            ymax = yv > ymax ? yv : ymax;
        }
    }
    if ((unsigned __int128)((__int128)xmax - xmin) > 0xfffffffeull || (unsigned __int128)((__int128)ymax - ymin) > 0xfffffffeull) {
        tdt_set_error("tdt_dbscan: coordinate span >= 2^32 is outside the device path's domain");
        return TDT_E_UNSUPPORTED;
    }
    void *h = nullptr, *d = nullptr;
    int rc = tdt_pinned(ctx, 1, n * 8 + n * 8 + 64, &h);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 5, n * 8 + n * 8 + 64, &d);
    if (rc) return rc;
    uint32_t *hx = (uint32_t *)h, *hy = hx + n;
    for (size_t i = 0; i < n; i++) {
        hx[i] = (uint32_t)(data[i * stride] - xmin);
        hy[i] = stride >= 2 ? (uint32_t)(data[i * stride + 1] - ymin) : 0u;
    }
    uint32_t *dx = (uint32_t *)d, *dy = dx + n;
    TDT_HIP(hipMemcpyAsync(dx, hx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    const int64_t boff[2] = {0, (int64_t)n};
    void *dlast = nullptr;
    rc = tdt_scratch(ctx, 6, n * 8 + 64, &dlast);
    if (rc) return rc;
    double *dl = (double *)dlast;
    long long *dlid = (long long *)((char *)dlast + n * 8);
    rc = tdt_dbscan_device(ctx, dx, dy, n, boff, 1, db_eps_u64(eps), m, mode, dl, (int64_t *)dlid);
    if (rc) return rc;
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    long long lid = -1;
    TDT_HIP(hipMemcpy(labels, dl, n * 8, hipMemcpyDeviceToHost));
    TDT_HIP(hipMemcpy(&lid, dlid, 8, hipMemcpyDeviceToHost));
    if (last_id) *last_id = lid;
    return TDT_OK;
}

// ---- y pass on caller-supplied x labels (DBSCAN.y_coordinate_clustering, DBSCAN.py:66-123) -------------------------------------
// d_xlab: int32 labels, -1 = unlabelled, every label value one contiguous index range, values ascending along the array (what
// x_coordinate_clustering returns, for any eps / m).  Sub-run 1 of a cluster keeps its label, extra sub-runs get cluster_id + 1,
// cluster_id + 2, ... in cluster order; *d_last_id = the final cluster_id.  *too_large != 0: a cluster has more than DB_SMALL
// members (or m > 64) — the labels were NOT produced; the caller takes another route.
extern "C" int tdt_dbscan_y_device(tdt_ctx *ctx, const int32_t *d_xlab, const uint32_t *d_y, size_t n_, uint64_t eps, int m, int64_t cluster_id,
                                   double *d_labels, int64_t *d_last_id, int *too_large) {
    if (!ctx || !too_large || m < 2 || (n_ && (!d_xlab || !d_y || !d_labels))) {
        tdt_set_error("tdt_dbscan_y_device: bad argument");
        return TDT_E_ARG;
    }
    *too_large = 0;
    if (n_ >= 0x7fff0000ull || m > DBF_M_MAX) {
        *too_large = 1;
        return TDT_OK;
    }
    if (n_ == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n = (int)n_;
    const int ntt = (n + DT_T - 1) / DT_T;
    const size_t sz_flags = 256, sz_agg = db_align((size_t)ntt * 4), sz_grp = db_align((size_t)2 * DT_GRPMAX * 4);
    void *ts = nullptr, *tb = nullptr;
    int rc = tdt_scratch(ctx, 20, sz_flags + 2 * sz_grp, &ts);
    if (rc) return rc;
    const size_t sz_code = db_align((size_t)n * 2 + 16);
    rc = tdt_scratch(ctx, 21, 2 * sz_agg + 1024 + sz_code, &tb);
    if (rc) return rc;
    unsigned short *t_code = (unsigned short *)((char *)tb + 2 * sz_agg + 1024);
    unsigned *t_flags = (unsigned *)ts;
    unsigned *t_grp0 = (unsigned *)((char *)ts + sz_flags), *t_grp1 = (unsigned *)((char *)ts + sz_flags + sz_grp);
    unsigned *t_aggR = (unsigned *)tb, *t_aggE = (unsigned *)((char *)tb + sz_agg);
    if (ctx->tile_flags_zeroed != ts) {
        TDT_HIP(hipMemsetAsync(t_flags, 0, sz_flags, st));
        TDT_HIP(hipMemsetAsync(t_grp0, 0, 2 * sz_grp, st));
        ctx->tile_flags_zeroed = ts;
    }
    DtParams TP;
    TP.x = (const unsigned *)d_xlab;
    TP.y = d_y;
    TP.n = n;
    TP.boff = nullptr;
    TP.nb = 1;
    TP.eps32 = eps > 0xffffffffull ? 0xffffffffu : (unsigned)eps;
    TP.wide = eps > 0xffffffffull;
    TP.m = m;
    TP.code = t_code;
    TP.aggR = t_aggR;
    TP.aggE = t_aggE;
    TP.brun = TP.bext = nullptr;
    TP.flags = t_flags;
    const bool odd = (ctx->tile_calls++ & 1u) != 0;
    TP.grp = odd ? t_grp1 : t_grp0;
    void *hp = nullptr;
    rc = tdt_pinned(ctx, 2, 64, &hp);
    if (rc) return rc;
    volatile unsigned *hw = (volatile unsigned *)hp;
    static std::atomic<unsigned> y_seq{0x40000000u};
    const unsigned seq = ++y_seq | 0x40000000u;
    hw[1] = 0;
    hipLaunchKernelGGL((dbt_tile<true, false, true>), dim3(dbt_grid(ctx, ntt)), dim3(DT_THREADS), 0, st, TP);
    ctx->tile_groups_max = std::max(ctx->tile_groups_max, (ntt + DT_GRP - 1) / DT_GRP);
    hipLaunchKernelGGL(dbt_finish1, dim3((ntt + DT_FTPB - 1) / DT_FTPB), dim3(256), 0, st, (const unsigned short *)t_code, d_labels, n, (const int *)d_xlab,
                       (const unsigned *)t_aggR, (const unsigned *)t_aggE, ntt, (const unsigned *)TP.grp, odd ? t_grp0 : t_grp1, ctx->tile_groups_max,
                       (long long *)d_last_id, (long long)cluster_id, 1, t_flags, hw, seq);
    TDT_CHECK_LAUNCH();
    bool seen = false;
    for (long spin = 0; spin < 4000000; spin++) {
        if (hw[1] == seq) {
            seen = true;
            break;
        }
        __builtin_ia32_pause();
    }
    if (!seen) TDT_HIP(hipStreamSynchronize(st));
    if (hw[0]) *too_large = 1;
    return TDT_OK;
}

extern "C" int tdt_dbscan_y(tdt_ctx *ctx, const int64_t *data, size_t n, size_t stride, double eps, int m, int64_t cluster_id, double *labels,
                            int64_t *last_id) {
    if (!ctx || stride < 2 || (n && (!data || !labels))) {
        tdt_set_error("tdt_dbscan_y: bad argument");
        return TDT_E_ARG;
    }
    if (m < 2) {
        tdt_set_error("tdt_dbscan_y: m must be >= 2");
        return TDT_E_ARG;
    }
    if (last_id) *last_id = cluster_id;
    if (n == 0) return TDT_OK;
    if (n >= 0x7fff0000ull) {
        tdt_set_error("tdt_dbscan_y: n too large");
        return TDT_E_UNSUPPORTED;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    // the labels this path takes: integers >= -1, every value one contiguous range, 0, 1, 2, ... along the array (the reference visits
    // `set(clusters)` — ascending for such values — and selects members by value; other label arrays are not reproduced here)
    long long next = 0, cur = -2;
    for (size_t i = 0; i < n; i++) {
        const double v = labels[i];
        const long long l = (long long)v;
        if ((double)l != v || l < -1 || l > 0x7ffffff0ll) {
            tdt_set_error("tdt_dbscan_y: label %g at %zu is not an integer in [-1, 2^31)", v, i);
            return TDT_E_UNSUPPORTED;
        }
        if (l == cur) continue;
        if (l >= 0) {
            if (l != next) {
                tdt_set_error("tdt_dbscan_y: labels must number contiguous clusters 0, 1, 2, ... along the array (label %lld at %zu, expected %lld)", l, i, next);
                return TDT_E_UNSUPPORTED;
            }
            next++;
        }
        cur = l;
    }
    if (cluster_id < next - 1) {
        // extra sub-runs would be numbered cluster_id + k <= the largest label: the reference's later `clusters == cluster` masks pick
        // them up again (DBSCAN.py:72,115) — not the closed form of this path
        tdt_set_error("tdt_dbscan_y: cluster_id %lld is below the largest label %lld (ids would collide with clusters not visited yet)",
                      (long long)cluster_id, next - 1);
        return TDT_E_UNSUPPORTED;
    }
    int64_t ymin = data[1], ymax = data[1];
    for (size_t i = 0; i < n; i++) {
        ymin = std::min(ymin, data[i * stride + 1]);
        ymax = std::max(ymax, data[i * stride + 1]);
    }
    if ((unsigned __int128)((__int128)ymax - ymin) > 0xfffffffeull) {
        tdt_set_error("tdt_dbscan_y: coordinate span >= 2^32 is outside the device path's domain");
        return TDT_E_UNSUPPORTED;
    }
    void *h = nullptr, *d = nullptr;
    int rc = tdt_pinned(ctx, 1, n * 8 + 64, &h);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 5, n * 16 + 256, &d);
    if (rc) return rc;
    uint32_t *hy = (uint32_t *)h;
    int32_t *hl = (int32_t *)(hy + n);
    for (size_t i = 0; i < n; i++) {
        hy[i] = (uint32_t)(data[i * stride + 1] - ymin);
        hl[i] = (int32_t)labels[i];
    }
    uint32_t *dy = (uint32_t *)d;
    int32_t *dl = (int32_t *)(dy + n);
    double *dlab = (double *)((char *)d + ((n * 8 + 255) & ~(size_t)255));
    hipStream_t st = ctx->stream;
    TDT_HIP(hipMemcpyAsync(dy, hy, n * 8, hipMemcpyHostToDevice, st));
    void *dlast = nullptr;
    rc = tdt_scratch(ctx, 6, 64, &dlast);
    if (rc) return rc;
    int large = 0;
    rc = tdt_dbscan_y_device(ctx, dl, dy, n, db_eps_u64(eps), m, cluster_id, dlab, (int64_t *)dlast, &large);
    if (rc) return rc;
    if (large) {
        tdt_set_error("tdt_dbscan_y: an x-cluster has more than %d members (or m > %d): not on the caller-supplied-labels path", DB_SMALL, DBF_M_MAX);
        return TDT_E_UNSUPPORTED;
    }
    TDT_HIP(hipStreamSynchronize(st));
    long long lid = cluster_id;
    TDT_HIP(hipMemcpy(labels, dlab, n * 8, hipMemcpyDeviceToHost));
    TDT_HIP(hipMemcpy(&lid, dlast, 8, hipMemcpyDeviceToHost));
    if (last_id) *last_id = lid;
    return TDT_OK;
}

// -------------------------------------------------------------------- sort + cluster in one call
extern "C" int tdt_sort_dbscan_ex(tdt_ctx *ctx, const int64_t *posA, const int64_t *posB, size_t n, const int64_t *bucket_off, int nb,
                                  double eps, int m, uint32_t *perm_out, double *labels_out, int64_t *runs_out, int64_t *last_out);
__global__ __launch_bounds__(DB_THREADS) void sd_make_keys(const unsigned *__restrict__ x, int n, const int *__restrict__ boff, int nb,
                                                           unsigned long long *__restrict__ key, unsigned *__restrict__ val) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    const int w0 = i & ~63;                                        // the wave's 64 consecutive positions
    if (i >= n) return;
    key[i] = ((unsigned long long)(unsigned)db_bucket_wave(boff, nb, w0, min(w0 + 63, n - 1), i) << 32) | x[i];   // stable sort => ties keep signal order
    val[i] = (unsigned)i;
}

__global__ __launch_bounds__(DB_THREADS) void sd_unpack(const unsigned long long *__restrict__ ksorted, const unsigned *__restrict__ vsorted,
                                                        const unsigned *__restrict__ y, int n, unsigned *__restrict__ xs,
                                                        unsigned *__restrict__ ysrt, unsigned *__restrict__ perm) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const unsigned src = vsorted[i];
    xs[i] = (unsigned)ksorted[i];
    ysrt[i] = y[src];
    perm[i] = src;
}

// labels on the wire: -1 or an id below 2^31, so 4 bytes each cross PCIe and the host widens them again (exact)
__global__ __launch_bounds__(DB_THREADS) void sd_labels_i32(const double *__restrict__ lab, int n, int *__restrict__ out) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i < n) out[i] = (int)lab[i];
}

extern "C" int tdt_sort_dbscan(tdt_ctx *ctx, const int64_t *posA, const int64_t *posB, size_t n, const int64_t *bucket_off, int nb,
                               double eps, int m, uint32_t *perm_out, double *labels_out) {
    return tdt_sort_dbscan_ex(ctx, posA, posB, n, bucket_off, nb, eps, m, perm_out, labels_out, nullptr, nullptr);
}

// the same call, also reporting per bucket the number of x-runs (DBSCAN.py:33-64: the x pass's cluster_id + 1) and the final
// cluster_id of DBSCAN.main — what a caller that cut one bucket into pieces needs to re-base the pieces' ids (dist.py)
extern "C" int tdt_sort_dbscan_ex(tdt_ctx *ctx, const int64_t *posA, const int64_t *posB, size_t n, const int64_t *bucket_off, int nb,
                                  double eps, int m, uint32_t *perm_out, double *labels_out, int64_t *runs_out, int64_t *last_out) {
    if (!ctx || nb < 1 || !bucket_off || (n && (!posA || !posB || !perm_out || !labels_out))) {
        tdt_set_error("tdt_sort_dbscan: bad argument");
        return TDT_E_ARG;
    }
    if (m < 2) {
        tdt_set_error("tdt_sort_dbscan: m must be >= 2");
        return TDT_E_ARG;
    }
    if (n >= 0x7fffffffull || bucket_off[0] != 0 || bucket_off[nb] != (int64_t)n) {
        tdt_set_error("tdt_sort_dbscan: bad bucket offsets / n");
        return TDT_E_ARG;
    }
    if (n == 0) {
        for (int b = 0; b < nb; b++) {
            if (runs_out) runs_out[b] = 0;
            if (last_out) last_out[b] = -1;
        }
        return TDT_OK;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    // column ranges and the 32-bit offsets: host passes over n elements, spread over the host threads
    const int nth = (int)std::max<size_t>(1, std::min<size_t>((size_t)tdt_host_thread_count(), n / (1u << 16) + 1));
    auto par = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < nth; t++) th.emplace_back([&, t] { fn(t, n * (size_t)t / nth, n * (size_t)(t + 1) / nth); });
        fn(0, 0, n / nth);
        for (auto &x : th) x.join();
    };
    std::vector<int64_t> lo_a(nth, posA[0]), hi_a(nth, posA[0]), lo_b(nth, posB[0]), hi_b(nth, posB[0]);
    par([&](int t, size_t i0, size_t i1) {
        int64_t a0 = posA[0], a1 = posA[0], b0 = posB[0], b1 = posB[0];
        for (size_t i = i0; i < i1; i++) {
            a0 = std::min(a0, posA[i]);
            a1 = std::max(a1, posA[i]);
            b0 = std::min(b0, posB[i]);
            b1 = std::max(b1, posB[i]);
        }
        lo_a[t] = a0; hi_a[t] = a1; lo_b[t] = b0; hi_b[t] = b1;
    });
    const int64_t amin = *std::min_element(lo_a.begin(), lo_a.end()), amax = *std::max_element(hi_a.begin(), hi_a.end());
    const int64_t bmin = *std::min_element(lo_b.begin(), lo_b.end()), bmax = *std::max_element(hi_b.begin(), hi_b.end());
    if ((unsigned __int128)((__int128)amax - amin) > 0xfffffffeull || (unsigned __int128)((__int128)bmax - bmin) > 0xfffffffeull) {
        tdt_set_error("tdt_sort_dbscan: coordinate span >= 2^32 is outside the device path's domain");
        return TDT_E_UNSUPPORTED;
    }
    hipStream_t st = ctx->stream;
    void *h = nullptr, *d = nullptr;
    const size_t hb = n * 12 + (size_t)(nb + 1) * 4 + 64;     // in: two 32-bit columns; out: labels (8 B) + order (4 B) per signal
    int rc = tdt_pinned(ctx, 1, hb, &h);
    if (rc) return rc;
    // device: x,y (in), xs,ys (sorted), perm, keys in/out, labels, boff
    const size_t szN4 = db_align(n * 4), szN8 = db_align(n * 8);
    rc = tdt_scratch(ctx, 5, 7 * szN4 + 3 * szN8 + db_align((size_t)(nb + 1) * 4) + 256, &d);
    if (rc) return rc;
    char *p = (char *)d;
    unsigned *dx = (unsigned *)p; p += szN4;
    unsigned *dy = (unsigned *)p; p += szN4;
    unsigned *dxs = (unsigned *)p; p += szN4;
    unsigned *dys = (unsigned *)p; p += szN4;
    unsigned *dperm = (unsigned *)p; p += szN4;
    unsigned *dv0 = (unsigned *)p; p += szN4;
    unsigned *dv1 = (unsigned *)p; p += szN4;
    unsigned long long *dk = (unsigned long long *)p; p += szN8;
    unsigned long long *dks = (unsigned long long *)p; p += szN8;
    double *dlab = (double *)p; p += szN8;
    int *dboff = (int *)p;
    uint32_t *hx = (uint32_t *)h, *hy = hx + n;
    int *hboff = (int *)((char *)h + n * 12);
    par([&](int, size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            hx[i] = (uint32_t)(posA[i] - amin);
            hy[i] = (uint32_t)(posB[i] - bmin);
        }
    });
    for (int b = 0; b <= nb; b++) hboff[b] = (int)bucket_off[b];
    TDT_HIP(hipMemcpyAsync(dx, hx, n * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dy, hy, n * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(dboff, hboff, (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st));
    const int blocks = ((int)n + DB_THREADS - 1) / DB_THREADS;
    hipLaunchKernelGGL(sd_make_keys, dim3(blocks), dim3(DB_THREADS), 0, st, (const unsigned *)dx, (int)n, (const int *)dboff, nb, dk, dv0);
    TDT_CHECK_LAUNCH();
    // only the digits that can differ are sorted: the posA span and the bucket index
    unsigned long long mask = amax > amin ? ((1ull << tdt_ceil_log2_u64((uint64_t)(amax - amin) + 1)) - 1ull) : 0ull;
    if (nb > 1) mask |= ((1ull << tdt_ceil_log2_u64((uint64_t)nb)) - 1ull) << 32;
    unsigned long long *ks = nullptr;
    unsigned *vs = nullptr;
    rc = tdt_radix_sort_pairs(ctx, dk, dv0, dks, dv1, n, mask, &ks, &vs);
    if (rc) return rc;
    hipLaunchKernelGGL(sd_unpack, dim3(blocks), dim3(DB_THREADS), 0, st, (const unsigned long long *)ks, (const unsigned *)vs,
                       (const unsigned *)dy, (int)n, dxs, dys, dperm);
    TDT_CHECK_LAUNCH();
    long long *dcnt = nullptr;
    if (runs_out || last_out) {
        void *dc = nullptr;
        rc = tdt_scratch(ctx, 6, (size_t)nb * 16 + 64, &dc);
        if (rc) return rc;
        dcnt = (long long *)dc;
        if (runs_out) {                      // the x pass alone: its cluster_id is the number of x-runs - 1
            rc = tdt_dbscan_device(ctx, dxs, dys, n, bucket_off, nb, db_eps_u64(eps), m, 1, dlab, (int64_t *)dcnt);
            if (rc) return rc;
        }
    }
    rc = tdt_dbscan_device(ctx, dxs, dys, n, bucket_off, nb, db_eps_u64(eps), m, 0, dlab, dcnt ? (int64_t *)(dcnt + nb) : nullptr);
    if (rc) return rc;
    if (dcnt) {
        std::vector<long long> hc((size_t)nb * 2);
        TDT_HIP(hipMemcpyAsync(hc.data(), dcnt, (size_t)nb * 16, hipMemcpyDeviceToHost, st));
        TDT_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < nb; b++) {
            if (runs_out) runs_out[b] = hc[b] + 1;
            if (last_out) last_out[b] = hc[nb + b];
        }
    }
    // results come back through the pinned block (its input columns are consumed by now), labels as int32 (dv0 is free again), then go
    // to the caller's arrays on the host threads
    int *dlab32 = (int *)dv0;
    hipLaunchKernelGGL(sd_labels_i32, dim3(blocks), dim3(DB_THREADS), 0, st, (const double *)dlab, (int)n, dlab32);
    TDT_CHECK_LAUNCH();
    int *hlab = (int *)h;
    uint32_t *hperm = (uint32_t *)((char *)h + n * 4);
    TDT_HIP(hipMemcpyAsync(hlab, dlab32, n * 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipMemcpyAsync(hperm, dperm, n * 4, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    par([&](int, size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) labels_out[i] = (double)hlab[i];
        memcpy(perm_out + i0, hperm + i0, (i1 - i0) * 4);
    });
    return TDT_OK;
}


// ---- the same, without host passes: 32-bit columns in (straight from the parsed signal tables), int32 labels in SIGNAL order out.
// Keys are (bucket << 32 | posA biased to unsigned); only the digits that can differ are sorted (max_pos bounds posA, e.g. the longest
// contig).  The posB column travels on the copy stream while the posA digits are being sorted; labels are scattered back to signal
// order on the device, so 4 B/signal return.  Pinned caller memory (tdt_host_alloc) is read and written by DMA directly; pageable
// memory goes through the context's pinned block.
__global__ __launch_bounds__(DB_THREADS) void sc_make_keys(const int *__restrict__ a, int n, const int *__restrict__ boff, int nb,
                                                           unsigned long long *__restrict__ key, unsigned *__restrict__ val) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    const int w0 = i & ~63;                                        // the wave's 64 consecutive positions
    if (i >= n) return;
    key[i] = ((unsigned long long)(unsigned)db_bucket_wave(boff, nb, w0, min(w0 + 63, n - 1), i) << 32) | ((unsigned)a[i] ^ 0x80000000u);   // order of signed values
    val[i] = (unsigned)i;
}

__global__ __launch_bounds__(DB_THREADS) void sc_unpack(const unsigned long long *__restrict__ ksorted, const unsigned *__restrict__ vsorted,
                                                        const int *__restrict__ b, int n, unsigned *__restrict__ xs, unsigned *__restrict__ ysrt) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    xs[i] = (unsigned)ksorted[i];
    ysrt[i] = (unsigned)b[vsorted[i]] ^ 0x80000000u;
}

__global__ __launch_bounds__(DB_THREADS) void sc_scatter_labels(const double *__restrict__ lab, const unsigned *__restrict__ perm, int n,
                                                                int *__restrict__ out) {
    const int i = blockIdx.x * DB_THREADS + threadIdx.x;
    if (i < n) out[perm[i]] = (int)lab[i];
}

static bool sc_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

extern "C" int tdt_cluster_columns(tdt_ctx *ctx, const int32_t *posA, const int32_t *posB, size_t n, const int64_t *bucket_off, int nb,
                                   double eps, int m, int64_t max_pos, int32_t *labels_by_signal, int64_t *runs_out, int64_t *last_out) {
    if (!ctx || nb < 1 || !bucket_off || (n && (!posA || !posB || !labels_by_signal))) {
        tdt_set_error("tdt_cluster_columns: bad argument");
        return TDT_E_ARG;
    }
    if (m < 2) {
        tdt_set_error("tdt_cluster_columns: m must be >= 2");
        return TDT_E_ARG;
    }
    if (n >= 0x7fffffffull || bucket_off[0] != 0 || bucket_off[nb] != (int64_t)n) {
        tdt_set_error("tdt_cluster_columns: bad bucket offsets / n");
        return TDT_E_ARG;
    }
    if (n == 0) {
        for (int b = 0; b < nb; b++) {
            if (runs_out) runs_out[b] = 0;
            if (last_out) last_out[b] = -1;
        }
        return TDT_OK;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream, cs = ctx->copy_stream;
    static const bool cc_timing = getenv("TIDDIT_CC_TIMING") != nullptr;          // host clock at the call's seams, to stderr
    double tm[12];
    int ntm = 0;
    auto mark = [&]() {
        if (cc_timing && ntm < 12) tm[ntm++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    mark();
    const size_t szN4 = db_align(n * 4), szN8 = db_align(n * 8);
    void *d = nullptr;
    int rc = tdt_scratch(ctx, 5, 7 * szN4 + 3 * szN8 + db_align((size_t)(nb + 1) * 4) + db_align((size_t)nb * 16 + 64) + 256, &d);
    if (rc) return rc;
    char *p = (char *)d;
    int *da = (int *)p; p += szN4;
    int *db = (int *)p; p += szN4;
    unsigned *dxs = (unsigned *)p; p += szN4;
    unsigned *dys = (unsigned *)p; p += szN4;
    unsigned *dv0 = (unsigned *)p; p += szN4;
    unsigned *dv1 = (unsigned *)p; p += szN4;
    int *dlab32 = (int *)p; p += szN4;
    unsigned long long *dk = (unsigned long long *)p; p += szN8;
    unsigned long long *dks = (unsigned long long *)p; p += szN8;
    double *dlab = (double *)p; p += szN8;
    int *dboff = (int *)p; p += db_align((size_t)(nb + 1) * 4);
    long long *dcnt = (long long *)p;
    // staging only for pageable caller memory
    mark();
    const bool pin_in = sc_is_pinned(posA) && sc_is_pinned(posB), pin_out = sc_is_pinned(labels_by_signal);
    mark();
    void *h = nullptr;
    rc = tdt_pinned(ctx, 1, n * 8 + (size_t)(nb + 1) * 4 + 64, &h);
    if (rc) return rc;
    int *hboff = (int *)((char *)h + n * 8);
    TDT_HIP(hipStreamSynchronize(st));                       // an earlier call may still be using the pinned block / the scratch
    mark();
    for (int b = 0; b <= nb; b++) hboff[b] = (int)bucket_off[b];
    const int32_t *srcA = posA, *srcB = posB;
    if (!pin_in) {
        const int nth = (int)std::max<size_t>(1, std::min<size_t>((size_t)tdt_host_thread_count(), n / (1u << 18) + 1));
        std::vector<std::thread> th;
        auto part = [&](int t) {
            const size_t i0 = n * (size_t)t / nth, i1 = n * (size_t)(t + 1) / nth;
            memcpy((int *)h + i0, posA + i0, (i1 - i0) * 4);
            memcpy((int *)h + n + i0, posB + i0, (i1 - i0) * 4);
        };
        for (int t = 1; t < nth; t++) th.emplace_back(part, t);
        part(0);
        for (auto &x : th) x.join();
        srcA = (const int32_t *)h;
        srcB = (const int32_t *)h + n;
    }
    TDT_HIP(hipMemcpyAsync(dboff, hboff, (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemcpyAsync(da, srcA, n * 4, hipMemcpyHostToDevice, st));
    // posB rides along with the sort of the posA digits — but only once posA has crossed: two copies in the same direction share the
    // link, and the sort waits for posA alone (started together both took 0.70 ms per 20 MB; in sequence posA is there after 0.36)
    TDT_HIP(hipEventRecord(ctx->ev[2], st));
    TDT_HIP(hipStreamWaitEvent(cs, ctx->ev[2], 0));
    TDT_HIP(hipMemcpyAsync(db, srcB, n * 4, hipMemcpyHostToDevice, cs));
    TDT_HIP(hipEventRecord(ctx->ev[3], cs));
    mark();
    const int blocks = ((int)n + DB_THREADS - 1) / DB_THREADS;
    hipLaunchKernelGGL(sc_make_keys, dim3(blocks), dim3(DB_THREADS), 0, st, (const int *)da, (int)n, (const int *)dboff, nb, dk, dv0);
    TDT_CHECK_LAUNCH();
    const uint64_t span = max_pos > 0 && max_pos < 0x7fffffffll ? (uint64_t)max_pos + 1 : 0x80000000ull;
    // biased keys: non-negative positions are 0x80000000 + pos, so the top bit is constant and only the span's digits differ
    unsigned long long mask = (1ull << tdt_ceil_log2_u64(span)) - 1ull;
    if (max_pos <= 0) mask = 0xffffffffull;                  // no bound given: all 32 bits (negative values included)
    if (nb > 1) mask |= ((1ull << tdt_ceil_log2_u64((uint64_t)nb)) - 1ull) << 32;
    unsigned long long *ks = nullptr;
    unsigned *vs = nullptr;
    rc = tdt_radix_sort_pairs(ctx, dk, dv0, dks, dv1, n, mask, &ks, &vs);
    if (rc) return rc;
    mark();
    TDT_HIP(hipStreamWaitEvent(st, ctx->ev[3], 0));
    hipLaunchKernelGGL(sc_unpack, dim3(blocks), dim3(DB_THREADS), 0, st, (const unsigned long long *)ks, (const unsigned *)vs, (const int *)db, (int)n,
                       dxs, dys);
    TDT_CHECK_LAUNCH();
    if (runs_out) {
        rc = tdt_dbscan_device(ctx, dxs, dys, n, bucket_off, nb, db_eps_u64(eps), m, 1, dlab, (int64_t *)dcnt);
        if (rc) return rc;
    }
    rc = tdt_dbscan_device(ctx, dxs, dys, n, bucket_off, nb, db_eps_u64(eps), m, 0, dlab, (runs_out || last_out) ? (int64_t *)(dcnt + nb) : nullptr);
    if (rc) return rc;
    mark();
    hipLaunchKernelGGL(sc_scatter_labels, dim3(blocks), dim3(DB_THREADS), 0, st, (const double *)dlab, (const unsigned *)vs, (int)n, dlab32);
    TDT_CHECK_LAUNCH();
    int *dst = pin_out ? labels_by_signal : (int *)h;
    TDT_HIP(hipMemcpyAsync(dst, dlab32, n * 4, hipMemcpyDeviceToHost, st));
    std::vector<long long> hc((size_t)nb * 2);
    if (runs_out || last_out) TDT_HIP(hipMemcpyAsync(hc.data(), dcnt, (size_t)nb * 16, hipMemcpyDeviceToHost, st));
    mark();
    TDT_HIP(hipStreamSynchronize(st));
    mark();
    if (!pin_out) memcpy(labels_by_signal, h, n * 4);
    for (int b = 0; b < nb; b++) {
        if (runs_out) runs_out[b] = hc[b] + 1;
        if (last_out) last_out[b] = hc[nb + b];
    }
    if (cc_timing) {
        fprintf(stderr, "tdt_cluster_columns n=%zu nb=%d us:", n, nb);
        for (int i = 1; i < ntm; i++) fprintf(stderr, " %.0f", tm[i] - tm[i - 1]);
        fprintf(stderr, "  (scratch | pinned? | pinned block + sync | copies issued | keys + sort issued | unpack + clustering | scatter + D2H issued | wait)\n");
    }
    return TDT_OK;
}

// pinned host memory for callers that build their columns in place (numpy arrays over it: tiddit_amd/hostutil.py)
extern "C" int tdt_host_alloc(size_t bytes, void **out) {
    if (!out) return TDT_E_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        tdt_set_error("tdt_host_alloc: pinned allocation of %zu bytes failed", bytes);
        return TDT_E_NOMEM;
    }
    return TDT_OK;
}
extern "C" int tdt_host_free(void *p) {
    if (p) (void)hipHostFree(p);
    return TDT_OK;
}

#ifdef DT_PROF
// variant builds only: per-phase cycles of dbt_tile summed over the workgroups of the last launch
extern "C" int tdt_debug_dt_prof(unsigned long long *out16) {
    static unsigned h[DT_PROF_TILES * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dt_prof), sizeof(h)) != hipSuccess) return -1;
    for (int k = 0; k < 16; k++) out16[k] = 0;
    for (int t = 0; t < DT_PROF_TILES; t++)
        for (int k = 0; k < 16; k++) out16[k] += h[t * 16 + k];
    return 0;
}
#endif
