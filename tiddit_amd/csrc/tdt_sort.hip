// Stable LSD radix sort of (uint64 key, uint32 value) pairs for gfx950 — hand-written, no library.
//
// Used for (i) the stable sort by posA of every (chrA,chrB) bucket (tiddit_cluster.pyx:152:
// key = bucket << 32 | posA, value = signal index) and (ii) the stable sort by posB inside x-clusters too
// large for the in-kernel rank sort (DBSCAN.py:79-81: key = cluster << 32 | posB, value = position).
// One pass per 8-bit digit that actually varies:
//   rs_hist     256-bin digit histogram of every 4096-key tile (LDS atomics), stored digit-major
//   scan        inclusive scan of the 256 x ntiles counts (the clustering path's scan kernels)
//   rs_scatter  every wave owns a contiguous quarter of the tile and walks it 64 keys at a time; lanes with
//               the same digit are found with 8 ballots (wavefront match), ranked by popcount below the
//               lane and placed after the running per-wave/per-digit counter kept in LDS => stable.
// HBM traffic per pass: 8 B read (hist) + 12 B read + 12 B written (scatter) per pair.
#include "tdt_common.h"

#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)
#define RS_ROUNDS 16
#define RS_TILE (RS_THREADS * RS_ROUNDS)  // 4096 keys per workgroup

int tdt_scan_u32_inclusive(tdt_ctx *ctx, unsigned *d_v, int n, unsigned *d_tsum);  // tdt_dbscan.hip

typedef unsigned long long ull;

__global__ __launch_bounds__(RS_THREADS) void rs_hist(const ull *__restrict__ keys, int n, int shift, int ntiles,
                                                      unsigned *__restrict__ hist) {
    __shared__ unsigned h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = t0 + r * RS_THREADS + tid;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * ntiles + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter(const ull *__restrict__ keys, const unsigned *__restrict__ vals, int n,
                                                         int shift, int ntiles, const unsigned *__restrict__ incl,
                                                         ull *__restrict__ keys_out, unsigned *__restrict__ vals_out) {
    __shared__ unsigned wh[RS_WAVES][256];   // per-wave digit counts, then running destinations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = 0; w < RS_WAVES; w++) wh[w][tid] = 0;
    __syncthreads();
    const int w0 = blockIdx.x * RS_TILE + wave * (RS_TILE / RS_WAVES);   // this wave's contiguous 1024 keys
    ull k[RS_ROUNDS];
    unsigned v[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        k[r] = i < n ? keys[i] : 0ull;
        v[r] = i < n ? vals[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        if (i < n) atomicAdd(&wh[wave][(unsigned)(k[r] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // digit `tid`: first destination of every wave = keys of smaller digits + earlier tiles + earlier waves
        const size_t idx = (size_t)tid * ntiles + blockIdx.x;
        unsigned base = idx ? incl[idx - 1] : 0u;
        for (int w = 0; w < RS_WAVES; w++) {
            const unsigned c = wh[w][tid];
            wh[w][tid] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        const bool valid = i < n;
        const unsigned d = (unsigned)(k[r] >> shift) & 255u;
        // lanes holding the same digit (wavefront match over the 8 digit bits)
        ull same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const ull bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        if (valid) {
            const unsigned below = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
            const unsigned dst = wh[wave][d] + below;
            keys_out[dst] = k[r];
            vals_out[dst] = v[r];
            if (below == 0) wh[wave][d] += (unsigned)__popcll(same);   // group leader advances the running counter
        }
        // the next round reads the counters this round wrote: same wave, LDS ops retire in order
        __builtin_amdgcn_wave_barrier();
    }
}

// Sort n pairs by the key bits selected in `bitmask` (only 8-bit digits containing a set bit are processed).
// keys/vals and the *_tmp buffers ping-pong; the final location is returned through out_keys/out_vals.
int tdt_radix_sort_pairs(tdt_ctx *ctx, ull *keys, unsigned *vals, ull *keys_tmp, unsigned *vals_tmp, size_t n_, ull bitmask,
                         ull **out_keys, unsigned **out_vals) {
    *out_keys = keys;
    *out_vals = vals;
    if (n_ == 0) return TDT_OK;
    const int n = (int)n_;
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    const int hn = 256 * ntiles;
    void *d_hist = nullptr, *d_ts = nullptr;
    int rc = tdt_scratch(ctx, 9, (size_t)hn * 4, &d_hist);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 10, ((size_t)hn / 1024 + 2) * 4, &d_ts);
    if (rc) return rc;
    ull *src_k = keys, *dst_k = keys_tmp;
    unsigned *src_v = vals, *dst_v = vals_tmp;
    for (int shift = 0; shift < 64; shift += 8) {
        if (!((bitmask >> shift) & 0xffull)) continue;
        hipLaunchKernelGGL(rs_hist, dim3(ntiles), dim3(RS_THREADS), 0, ctx->stream, (const ull *)src_k, n, shift, ntiles, (unsigned *)d_hist);
        TDT_CHECK_LAUNCH();
        rc = tdt_scan_u32_inclusive(ctx, (unsigned *)d_hist, hn, (unsigned *)d_ts);
        if (rc) return rc;
        hipLaunchKernelGGL(rs_scatter, dim3(ntiles), dim3(RS_THREADS), 0, ctx->stream, (const ull *)src_k, (const unsigned *)src_v, n, shift,
                           ntiles, (const unsigned *)d_hist, dst_k, dst_v);
        TDT_CHECK_LAUNCH();
        ull *tk = src_k; src_k = dst_k; dst_k = tk;
        unsigned *tv = src_v; src_v = dst_v; dst_v = tv;
    }
    *out_keys = src_k;
    *out_vals = src_v;
    return TDT_OK;
}
