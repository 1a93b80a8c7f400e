// Stable LSD radix sort of (uint64 key, uint32 value) pairs for gfx950 — hand-written, no library.
//
// Used for (i) the stable sort by posA of every (chrA,chrB) bucket (tiddit_cluster.pyx:152:
// key = bucket << 32 | posA, value = signal index) and (ii) the stable sort by posB inside x-clusters too
// large for the in-kernel rank sort (DBSCAN.py:79-81: key = cluster << 32 | posB, value = position).
//
// Round 6: a one-sweep sort.  (Rounds 1-5 ran, per 8-bit digit, a tile-histogram kernel that re-read the keys, three scan launches and a
// scatter whose lanes stored straight to their global destinations — up to 64 cache lines per store instruction: 5 launches and
// 32 B of traffic per pair and digit, 1.95 TB/s.)
//   rs_hist_all   ONE read of the keys builds the 256-bin histogram of EVERY digit the sort will process (LDS atomics, then one global
//                 atomic per non-empty bin and workgroup); it also notes the key bits outside the sorted ones, which are equal in all keys.
//   rs_onesweep   one launch per digit.  A workgroup takes the next tile of 4096 pairs off a counter (tiles are handed out in order, so
//                 every tile in front of it is already running), each wave ranks its contiguous quarter of the tile 64 keys at a time
//                 (lanes with the same digit are found with 8 ballots and ranked by popcount below the lane => stable), the tile's
//                 256 digit counts are PUBLISHED in a status word per digit ({flag, count}: local count first, inclusive prefix once known)
//                 and thread d LOOKS BACK over the tiles in front for the prefix of digit d (decoupled look-back: no scan kernels, the keys
//                 are read once per digit).  The pairs are then reordered by digit IN LDS and leave in tile order: consecutive lanes write
//                 consecutive addresses of a digit's run (a tile holds 16 pairs per digit on average: 128-byte runs of keys).
// A digit is up to 8 key bits taken from at most two bit fields (the posA span's last bits and the bucket index share a digit), so a
// 28-bit position span and 300 buckets take 5 launches, one bucket of 31-bit positions 4.  When no sorted bit lies above bit 31 the
// keys travel as 32-bit words between the first and the last digit (8 instead of 12 bytes per pair and direction).
// HBM traffic per pair: 8 B (histograms) + per digit 12 B read + 12 B written (8 + 8 for 32-bit keys).
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_sort;
const char *const tdt_variant_sort = ""
#ifdef RS_ROUNDS
    " RS_ROUNDS"
#endif
#ifdef RS_LOOK
    " RS_LOOK"
#endif
    ;

#include "tdt_common.h"

#include <algorithm>

#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / 64)
#ifndef RS_ROUNDS
#define RS_ROUNDS 16
#endif
#define RS_TILE (RS_THREADS * RS_ROUNDS)  // 4096 pairs per workgroup
#define RS_MAXPASS 8
#ifndef RS_LOOK
#define RS_LOOK 8                         // status words of the tiles in front fetched per round trip of the look-back
#endif
#define RS_FLAG_AGG 0x40000000u           // status word: the tile's own count of the digit
#define RS_FLAG_PREFIX 0x80000000u        // ... the count of the digit in this tile and every tile in front of it
#define RS_VALUE 0x3fffffffu

typedef unsigned long long ull;

struct RsPass {
    int s1, w1, s2, w2;                   // digit = bits [s1, s1 + w1) | bits [s2, s2 + w2) << w1
};
struct RsPlan {
    int np;
    RsPass p[RS_MAXPASS];
};

template <typename K>
__device__ __forceinline__ unsigned rs_digit(K k, const RsPass &P) {
    unsigned d = (unsigned)(k >> P.s1) & ((1u << P.w1) - 1u);
    if (P.w2) d |= ((unsigned)((ull)k >> P.s2) & ((1u << P.w2) - 1u)) << P.w1;
    return d;
}

// header of the workspace: ghist[RS_MAXPASS][256], then 16 words {tile counters [0..8), high word of the keys [8]}
#define RS_HDR_WORDS (RS_MAXPASS * 256 + 16)

__global__ __launch_bounds__(RS_THREADS) void rs_hist_all(const ull *__restrict__ keys, int n, RsPlan plan, unsigned *__restrict__ hdr) {
    __shared__ unsigned h[RS_MAXPASS][256];
    const int tid = threadIdx.x;
    for (int p = 0; p < plan.np; p++) h[p][tid] = 0;
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) hdr[RS_MAXPASS * 256 + 8] = (unsigned)(keys[0] >> 32);
    // (16-byte loads: two keys per lane and trip; `keys` is 256-byte aligned scratch)
    const size_t pairs = (size_t)n / 2;
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + tid; i < pairs; i += (size_t)gridDim.x * RS_THREADS) {
        const ulonglong2 kk = reinterpret_cast<const ulonglong2 *>(keys)[i];
#pragma unroll
        for (int p = 0; p < RS_MAXPASS; p++)
            if (p < plan.np) {
                atomicAdd(&h[p][rs_digit(kk.x, plan.p[p])], 1u);
                atomicAdd(&h[p][rs_digit(kk.y, plan.p[p])], 1u);
            }
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const ull k = keys[n - 1];
        for (int p = 0; p < plan.np; p++) atomicAdd(&h[p][rs_digit(k, plan.p[p])], 1u);
    }
    __syncthreads();
    for (int p = 0; p < plan.np; p++) {
        const unsigned c = h[p][tid];
        if (c) atomicAdd(&hdr[p * 256 + tid], c);
    }
}

// exclusive scan of one value per thread over the workgroup's 256 threads (s: 4 words of LDS)
__device__ __forceinline__ unsigned rs_block_excl(unsigned v, unsigned *s, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s[wave] = inc;
    __syncthreads();
    unsigned base = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++)
        if (w < wave) base += s[w];
    __syncthreads();
    return base + inc - v;
}

template <typename KI, typename KO>
__global__ __launch_bounds__(RS_THREADS) void rs_onesweep(const KI *__restrict__ keys, const unsigned *__restrict__ vals, int n, RsPass P,
                                                          const unsigned *__restrict__ ghist, unsigned *__restrict__ status,
                                                          unsigned *__restrict__ ctl, int pass, KO *__restrict__ keys_out,
                                                          unsigned *__restrict__ vals_out, int *__restrict__ err) {
    __shared__ KI sk[RS_TILE];
    __shared__ unsigned sv[RS_TILE];
    __shared__ unsigned wh[RS_WAVES][256];   // per-wave running digit counts, then each wave's first slot inside the digit's run
    __shared__ int gdst[256];                // global destination of the tile's slot 0 if it held digit d  (destination = gdst[d] + slot)
    __shared__ unsigned sscan[RS_WAVES];
    __shared__ unsigned s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(&ctl[pass], 1u);
    for (int w = 0; w < RS_WAVES; w++) wh[w][tid] = 0;
    __syncthreads();
    const int tile = (int)s_tile;
    const int t0 = tile * RS_TILE;
    const int w0 = t0 + wave * (RS_TILE / RS_WAVES);   // this wave's contiguous 1024 pairs
    KI k[RS_ROUNDS];
    unsigned v[RS_ROUNDS];
    unsigned short rank[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        k[r] = i < n ? keys[i] : (KI)0;
        v[r] = i < n ? vals[i] : 0u;
    }
    // ---- rank inside the wave: digit groups by 8 ballots, the running count of every digit in LDS
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        const bool valid = i < n;
        const unsigned d = rs_digit(k[r], P);
        ull same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const ull bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        if (valid) {
            const unsigned below = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
            const unsigned c = wh[wave][d];
            rank[r] = (unsigned short)(c + below);
            if (below == 0) wh[wave][d] = c + (unsigned)__popcll(same);   // group leader advances the running counter
        }
        // the next round reads the counters this round wrote: same wave, LDS ops retire in order
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- digit `tid`: the tile's count, published; the waves' first slots inside the digit's run
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) {
        const unsigned c = wh[w][tid];
        wh[w][tid] = total;
        total += c;
    }
    unsigned *const my = status + (size_t)tile * 256 + tid;
    __hip_atomic_store(my, (tile == 0 ? RS_FLAG_PREFIX : RS_FLAG_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned lbase = rs_block_excl(total, sscan, tid);           // where the digit's run starts in the reordered tile
    const unsigned gbase = rs_block_excl(ghist[tid], sscan, tid);      // ... and in the output: keys of smaller digits
    // the waves' first slots become tile slots
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) wh[w][tid] += lbase;
    __syncthreads();
    // ---- reorder by digit in LDS (the tiles in front publish their counts meanwhile)
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int i = w0 + r * 64 + lane;
        if (i < n) {
            const unsigned d = rs_digit(k[r], P);
            const unsigned slot = wh[wave][d] + rank[r];
            sk[slot] = k[r];
            sv[slot] = v[r];
        }
    }
    // ---- look back over the tiles in front for the digit's prefix, RS_LOOK status words per round trip (one word at a time, a tile whose
    // 500 predecessors were started together and hold only their own counts walked them one memory latency each)
    unsigned excl = 0;
    for (int t = tile - 1; t >= 0; t -= RS_LOOK) {
        unsigned s[RS_LOOK];
#pragma unroll
        for (int j = 0; j < RS_LOOK; j++)
            s[j] = t - j >= 0 ? __hip_atomic_load(status + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RS_FLAG_PREFIX;
        bool done = false;
#pragma unroll
        for (int j = 0; j < RS_LOOK; j++) {
            if (done) continue;
            unsigned sj = s[j];
            unsigned spins = 0;
            while (!(sj & ~RS_VALUE)) {                                // the tile is running (it took its number before this one): wait for its count
                __builtin_amdgcn_s_sleep(1);
                sj = __hip_atomic_load(status + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1u << 24)) {                            // (cannot happen; a hang would cost the box — report instead)
                    *err = 7;                                          // (tdt_ctx_sync reports it)
                    break;
                }
            }
            excl += sj & RS_VALUE;
            done = (sj & RS_FLAG_PREFIX) != 0;
        }
        if (done) break;
    }
    if (tile) __hip_atomic_store(my, RS_FLAG_PREFIX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gdst[tid] = (int)(gbase + excl) - (int)lbase;
    __syncthreads();
    // ---- and out, in tile order: consecutive lanes write consecutive addresses of a digit's run
    const int cnt = n - t0 < RS_TILE ? n - t0 : RS_TILE;
    const unsigned hi = sizeof(KO) > sizeof(KI) ? ctl[8] : 0u;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int slot = r * RS_THREADS + tid;
        if (slot < cnt) {
            const KI kk = sk[slot];
            const int dst = gdst[rs_digit(kk, P)] + slot;
            keys_out[dst] = sizeof(KO) > sizeof(KI) ? (KO)(((ull)hi << 32) | (ull)kk) : (KO)kk;
            vals_out[dst] = sv[slot];
        }
    }
}

// the digits of a sort: the set bits of `bitmask` from the lowest up, 8 per digit, a digit spanning at most two runs of set bits
static RsPlan rs_plan(ull bitmask) {
    RsPlan plan{};
    auto set = [&](int b) { return b < 64 && ((bitmask >> b) & 1ull); };
    auto more = [&](int b) { return b < 64 && (bitmask >> b) != 0; };
    int bit = 0;
    while (more(bit) && plan.np < RS_MAXPASS) {
        RsPass P{0, 0, 0, 0};
        while (!set(bit)) bit++;
        P.s1 = bit;
        while (set(bit) && P.w1 < 8) bit++, P.w1++;
        if (P.w1 < 8 && more(bit)) {
            while (!set(bit)) bit++;
            P.s2 = bit;
            while (set(bit) && P.w1 + P.w2 < 8) bit++, P.w2++;
        }
        plan.p[plan.np++] = P;
    }
    return plan;
}

// Sort n pairs by the key bits selected in `bitmask`; the key bits outside it must be equal in all keys (they are: callers pass the
// mask of the bits that can differ).  keys/vals and the *_tmp buffers ping-pong; the final location is returned through out_keys/out_vals.
int tdt_radix_sort_pairs(tdt_ctx *ctx, ull *keys, unsigned *vals, ull *keys_tmp, unsigned *vals_tmp, size_t n_, ull bitmask,
                         ull **out_keys, unsigned **out_vals) {
    *out_keys = keys;
    *out_vals = vals;
    if (n_ < 2 || !bitmask) return TDT_OK;
    if (n_ >= (1ull << 30)) {
        tdt_set_error("tdt_radix_sort_pairs: %zu pairs; the sort handles fewer than 2^30", n_);
        return TDT_E_RANGE;
    }
    const int n = (int)n_;
    RsPlan plan = rs_plan(bitmask);
    {   // more than 64 bits' worth of digits cannot happen; more than RS_MAXPASS digits only for masks with many separate runs
        ull covered = 0;
        for (int p = 0; p < plan.np; p++) {
            covered |= (((1ull << plan.p[p].w1) - 1ull) << plan.p[p].s1);
            if (plan.p[p].w2) covered |= (((1ull << plan.p[p].w2) - 1ull) << plan.p[p].s2);
        }
        if (covered != bitmask) {
            tdt_set_error("tdt_radix_sort_pairs: the key mask %llx needs more than %d digits", bitmask, RS_MAXPASS);
            return TDT_E_RANGE;
        }
    }
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    const size_t words = RS_HDR_WORDS + (size_t)plan.np * ntiles * 256;
    void *d_ws = nullptr;
    int rc = tdt_scratch(ctx, 9, words * 4, &d_ws);
    if (rc) return rc;
    unsigned *hdr = (unsigned *)d_ws, *ctl = hdr + RS_MAXPASS * 256, *status = hdr + RS_HDR_WORDS;
    hipStream_t st = ctx->stream;
    TDT_HIP(hipMemsetAsync(d_ws, 0, words * 4, st));
    const int hgrid = std::min(ntiles, 2 * ctx->num_cu);
    hipLaunchKernelGGL(rs_hist_all, dim3(hgrid), dim3(RS_THREADS), 0, st, (const ull *)keys, n, plan, hdr);
    TDT_CHECK_LAUNCH();
    const bool narrow = !(bitmask >> 32) && plan.np > 1;        // 32-bit keys between the first and the last digit
    void *src_k = keys, *dst_k = keys_tmp;
    unsigned *src_v = vals, *dst_v = vals_tmp;
    for (int p = 0; p < plan.np; p++) {
        const bool in64 = !narrow || p == 0, out64 = !narrow || p == plan.np - 1;
        unsigned *stp = status + (size_t)p * ntiles * 256;
        const unsigned *gh = hdr + p * 256;
        if (in64 && out64)
            hipLaunchKernelGGL((rs_onesweep<ull, ull>), dim3(ntiles), dim3(RS_THREADS), 0, st, (const ull *)src_k, (const unsigned *)src_v, n, plan.p[p], gh, stp, ctl, p, (ull *)dst_k, dst_v, ctx->d_async_err);
        else if (in64)
            hipLaunchKernelGGL((rs_onesweep<ull, unsigned>), dim3(ntiles), dim3(RS_THREADS), 0, st, (const ull *)src_k, (const unsigned *)src_v, n, plan.p[p], gh, stp, ctl, p, (unsigned *)dst_k, dst_v, ctx->d_async_err);
        else if (out64)
            hipLaunchKernelGGL((rs_onesweep<unsigned, ull>), dim3(ntiles), dim3(RS_THREADS), 0, st, (const unsigned *)src_k, (const unsigned *)src_v, n, plan.p[p], gh, stp, ctl, p, (ull *)dst_k, dst_v, ctx->d_async_err);
        else
            hipLaunchKernelGGL((rs_onesweep<unsigned, unsigned>), dim3(ntiles), dim3(RS_THREADS), 0, st, (const unsigned *)src_k, (const unsigned *)src_v, n, plan.p[p], gh, stp, ctl, p, (unsigned *)dst_k, dst_v, ctx->d_async_err);
        TDT_CHECK_LAUNCH();
        std::swap(src_k, dst_k);
        std::swap(src_v, dst_v);
    }
    *out_keys = (ull *)src_k;
    *out_vals = src_v;
    return TDT_OK;
}
