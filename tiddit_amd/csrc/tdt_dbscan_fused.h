// Ballot-mask kernels of the clustering path.  Included by tdt_dbscan.hip.  Used when m <= DBF_M_MAX; the
// multi-kernel byte-flag path in tdt_dbscan.hip stays as the general one.
//
// Every kernel walks tiles of 4096 points = 64 mask words (tile = workgroup index).  All per-point predicates are
// booleans, so the tile-local scans are done on 64-bit ballot masks: lane = point when predicates are
// evaluated (one coalesced load per array per 64 points), lane = word when the bit algebra is done; prefix
// counts are popcounts of masked ballots, run heads/tails are shifts of those masks — no per-thread item arrays,
// no LDS bank conflicts.  Cross-tile prefixes come from the per-tile aggregates of the previous launch.
// (An in-kernel chained scan with decoupled look-back over 8-byte status words was tried first and measured no
// faster than this launch-boundary form)
#pragma once

#define DBF_M_MAX 64
#ifndef DBF_THREADS
#define DBF_THREADS 256
#endif
#define DBF_WAVES (DBF_THREADS / 64)
#define DBF_STEPS (64 / DBF_WAVES)          // 64-element words per wave
#define DBF_WORDS (DBF_WAVES * DBF_STEPS)   // 64 words per tile
#define DBF_TILE (DBF_WORDS * 64)           // 4096 points

typedef unsigned long long ull;

struct DbfCtl {  // zeroed before every call
    unsigned ticket[4];
    unsigned nlarge;
    unsigned pad[3];
};

__device__ __forceinline__ ull dbf_le(int lane) { return lane >= 63 ? ~0ull : ((2ull << lane) - 1ull); }   // bits 0..lane
__device__ __forceinline__ ull dbf_lt(int lane) { return (1ull << lane) - 1ull; }                           // bits 0..lane-1
// any set bit among positions [i-m+1, i] of the bit stream (prev = the 64 positions before cur); 2 <= m <= 64
__device__ __forceinline__ ull dbf_smear(ull cur, ull prev, int m) {
    ull f = cur;
    for (int q = 1; q < m; q++) f |= (cur << q) | (prev >> (64 - q));
    return f;
}
__device__ __forceinline__ int dbf_popc(ull v) { return __popcll(v); }
// Mask words are wave-uniform.  Telling the compiler (readfirstlane) moves the 64-bit shift / popcount /
// find-first work from per-lane VALU (multi-cycle 64-bit ops on every lane) to the scalar unit.
__device__ __forceinline__ ull dbf_uni(ull v) {
    return ((ull)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
// number of set bits of a uniform mask at positions <= lane
__device__ __forceinline__ unsigned dbf_cnt_le(ull mask, int lane) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)) + (unsigned)((mask >> lane) & 1ull);
}

// ------------------------------------------------------------------------------------------ x pass
#define DBF_XSH (DBF_TILE + 128 + DBF_M_MAX)   // x staged for [t0-64, t0+TILE+64+m)

// p word: bit = p[g0 + lane]  (DBSCAN.py:41-51), 0 for points outside [0, n); x comes from the LDS stage.
// For m <= 4 (the reference's default is 3) the window is read with four independent LDS loads and masked,
// instead of a data-dependent loop that waits for every read in turn.
__device__ __forceinline__ ull dbf_px_word(const unsigned *xsh, int sh0, int n, const int *__restrict__ boff, int nb, ull eps, int m,
                                           int g0, int lane) {
    const int i = g0 + lane;
    bool p = false;
    if (i >= 0 && i < n) {
        const int bend = nb == 1 ? n : boff[db_bucket(boff, nb, i) + 1];
        if (i + m <= bend) {                          // `for i in range(0, len(data)-m+1)` (:39)
            const int hi = min(i + m, bend - 1);      // data[i+1:i+m+1] truncates at the array end (:43)
            const int o = i - sh0;
            const unsigned xi = xsh[o];
            unsigned maxd = 0;
            if (m <= 4) {
                const unsigned v1 = xsh[o + 1], v2 = xsh[o + 2], v3 = xsh[o + 3], v4 = xsh[o + 4];   // staged range covers o+4
                const int cnt = hi - i;               // 1..4 window members
                maxd = db_absdiff(v1, xi);
                if (cnt >= 2) maxd = max(maxd, db_absdiff(v2, xi));
                if (cnt >= 3) maxd = max(maxd, db_absdiff(v3, xi));
                if (cnt >= 4) maxd = max(maxd, db_absdiff(v4, xi));
            } else {
                for (int q = i + 1; q <= hi; q++) maxd = max(maxd, db_absdiff(xsh[q - sh0], xi));
            }
            p = (ull)maxd < eps;
        }
    }
    return __ballot(p);
}

__global__ void dbf_empty_buckets(const int *__restrict__ boff, int nb, long long *__restrict__ last_id) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb && boff[b] == boff[b + 1]) last_id[b] = -1;
}


// ==========================================================================================
// Launch-boundary variant: the same ballot-mask tiles, but the cross-tile prefix comes from a tiny
// single-workgroup scan between two launches instead of in-kernel look-back (no polling at all).
//   dbm_x_masks -> tile_scan -> dbm_x_labels ;  dbm_y_masks -> tile_scan -> dbm_y_mid -> tile_scan -> dbm_y_final
// Masks live in global memory: 8 B per 64 points per mask.
// ==========================================================================================
#define DBM_XSH4 ((DBF_TILE + 128 + DBF_M_MAX) / 4)   // uint4 chunks staged for [t0-64, t0+TILE+64+m)

__device__ __forceinline__ ull mono_pack(unsigned has, ull val) { return ((ull)(has & 1u) << 63) | (val & ~(1ull << 63)); }

// exclusive scan of the segmented-sum monoid over nt tile summaries, in place (one workgroup of 1024)
__global__ __launch_bounds__(1024) void tile_scan(ull *agg, int nt) {
    __shared__ unsigned wh[16];
    __shared__ ull wv[16];
    __shared__ unsigned c_has;
    __shared__ ull c_val;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        c_has = 0;
        c_val = 0;
    }
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
        const int i = base + tid;
        const ull w = i < nt ? agg[i] : 0ull;
        unsigned h = (unsigned)(w >> 63);
        ull v = w & ~(1ull << 63);
        const unsigned h0 = h;
        const ull v0 = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned oh = __shfl_up(h, d);
            const ull ov = __shfl_up(v, d);
            if (lane >= d) {
                if (!h) v += ov;
                h |= oh;
            }
        }
        if (lane == 63) {
            wh[wave] = h;
            wv[wave] = v;
        }
        __syncthreads();
        // state entering this thread = carry (+) earlier waves (+) earlier lanes of this wave
        unsigned eh = c_has;
        ull ev = c_val;
        for (int k = 0; k < wave; k++) {
            if (wh[k]) { eh = 1; ev = wv[k]; } else ev += wv[k];
        }
        unsigned ph = __shfl_up(h, 1);
        ull pv = __shfl_up(v, 1);
        if (lane == 0) { ph = 0; pv = 0; }
        if (ph) { eh = 1; ev = pv; } else ev += pv;
        if (i < nt) agg[i] = mono_pack(eh, ev);
        __syncthreads();
        if (tid == 1023) {   // inclusive state after this chunk
            unsigned th = eh;
            ull tv = ev;
            if (h0) { th = 1; tv = v0; } else tv += v0;
            c_has = th;
            c_val = tv;
        }
        __syncthreads();
    }
}

// Word-parallel mask math: a tile has exactly 64 mask words, so ONE wave does the per-word bit algebra
// with lane = word (64-bit VALU on 64 different words at once) instead of every wave repeating it
// word after word on uniform values; the per-point passes then only expand finished masks.

// inclusive->exclusive segmented scan over the 64 lanes of (has, val); cin = state entering lane 0
__device__ __forceinline__ void wave_seg_scan(unsigned has, unsigned val, unsigned cin, unsigned &enter, unsigned &tot_has,
                                              unsigned &tot_val) {
    const int lane = threadIdx.x & 63;
    unsigned h = has, v = val;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned oh = __shfl_up(h, d), ov = __shfl_up(v, d);
        if (lane >= d) {
            if (!h) v += ov;
            h |= oh;
        }
    }
    unsigned eh = __shfl_up(h, 1), ev = __shfl_up(v, 1);
    if (lane == 0) {
        eh = 0;
        ev = 0;
    }
    enter = eh ? ev : cin + ev;
    tot_has = __shfl(h, 63);
    tot_val = __shfl(v, 63);
}

// Exclusive state of `tile` = combine of agg[0..tile-1], computed by the consuming workgroup itself (no scan
// launch in between) when the tile count is small; `pre` (filled by tile_scan) is used otherwise.
#define DBM_INLINE_PREFIX_MAX 2048
__device__ __forceinline__ unsigned tile_prefix(const ull *__restrict__ agg, int tile, int ntiles, unsigned *shh, unsigned *shv) {
    if (ntiles > DBM_INLINE_PREFIX_MAX) return (unsigned)(agg[tile] & ~(1ull << 63));   // already scanned in place
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = (tile + DBF_THREADS - 1) / DBF_THREADS;
    unsigned h = 0, v = 0;
    for (int i = tid * cs; i < min(tile, (tid + 1) * cs); i++) {
        const ull w = agg[i];
        if (w >> 63) {
            h = 1;
            v = (unsigned)w;
        } else {
            v += (unsigned)w;
        }
    }
    unsigned enter, th, tv;
    wave_seg_scan(h, v, 0, enter, th, tv);
    if (lane == 0) {
        shh[wave] = th;
        shv[wave] = tv;
    }
    __syncthreads();
    unsigned r = 0;
    for (int w = 0; w < DBF_WAVES; w++) r = shh[w] ? shv[w] : r + shv[w];
    __syncthreads();
    return r;
}

__device__ __forceinline__ ull dbf_starts_x(ull cur, ull prev) { return cur & ~((cur << 1) | (prev >> 63)); }
__device__ __forceinline__ ull dbf_starts_y(ull cur, ull prev, ull h) { return cur & (h | ~((cur << 1) | (prev >> 63))); }
// set bits of `bits` at or after the highest set bit of `marks` (marks != 0)
__device__ __forceinline__ ull dbf_from_last(ull bits, ull marks) { return bits & (~0ull << (63 - __clzll((long long)marks))); }

// Branch-free p word for the common configuration (one bucket, m <= 4: the reference default is 3): four
// unconditional LDS reads (the staged range always covers o+4, zero filled past the array), selects instead
// of exec-mask branches.  eps32 = min(eps, 2^32-1); wide == eps > 2^32-1 (every 32-bit distance qualifies).
__device__ __forceinline__ ull dbf_px_word_fast(const unsigned *xsh, int sh0, int n, unsigned eps32, bool wide, int m, int g0, int lane) {
    const int i = g0 + lane;
    const int o = i - sh0;
    const unsigned xi = xsh[o], v1 = xsh[o + 1], v2 = xsh[o + 2], v3 = xsh[o + 3], v4 = xsh[o + 4];
    const int cnt = min(i + m, n - 1) - i;   // window members (data[i+1:i+m+1] truncated at the array end, :43)
    unsigned maxd = db_absdiff(v1, xi);
    maxd = cnt >= 2 ? max(maxd, db_absdiff(v2, xi)) : maxd;
    maxd = cnt >= 3 ? max(maxd, db_absdiff(v3, xi)) : maxd;
    maxd = cnt >= 4 ? max(maxd, db_absdiff(v4, xi)) : maxd;
    return __ballot(i >= 0 && i + m <= n && (wide || maxd < eps32));   // i <= n-m (:39)
}

// p masks of one tile (+ the count of run starts)  — x staged with 16-byte loads
template <bool FAST>
__global__ __launch_bounds__(DBF_THREADS) void dbm_x_masks(const unsigned *__restrict__ x, int n, const int *__restrict__ boff, int nb,
                                                           ull eps, int m, ull *__restrict__ PM, ull *__restrict__ agg,
                                                           ull *__restrict__ PY, DbfCtl *__restrict__ ctl) {
    __shared__ __attribute__((aligned(16))) unsigned xsh[DBM_XSH4 * 4];
    __shared__ ull pm[DBF_WORDS + 1];   // [0] = last word of the previous tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    if (tile == 0 && tid < 4) {   // the pass's few words of zero state (four memset launches otherwise): nothing reads them before
        if (tid == 0) PM[0] = 0;                                        // the word before the mask array ...
        if (tid == 1) PM[(size_t)gridDim.x * DBF_WORDS + 1] = 0;        // ... and the one after the last tile
        if (tid == 2) PY[0] = 0;
        if (tid == 3) *ctl = DbfCtl{};
    }
    const int t0 = tile * DBF_TILE;
    const int sh0 = t0 - 64;
    {   // all loads first (independent, one round trip), then the LDS stores
        constexpr int NCH = (DBM_XSH4 + DBF_THREADS - 1) / DBF_THREADS;
        uint4 v[NCH];
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int c = tid + k * DBF_THREADS;
            const int g = sh0 + 4 * c;
            if (c < DBM_XSH4 && g >= 0 && g + 4 <= n) v[k] = *reinterpret_cast<const uint4 *>(x + g);
            else if (c < DBM_XSH4) {
                v[k].x = (g >= 0 && g < n) ? x[g] : 0u;
                v[k].y = (g + 1 >= 0 && g + 1 < n) ? x[g + 1] : 0u;
                v[k].z = (g + 2 >= 0 && g + 2 < n) ? x[g + 2] : 0u;
                v[k].w = (g + 3 >= 0 && g + 3 < n) ? x[g + 3] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int c = tid + k * DBF_THREADS;
            if (c < DBM_XSH4) *reinterpret_cast<uint4 *>(xsh + 4 * c) = v[k];
        }
    }
    __syncthreads();
    const unsigned eps32 = eps > 0xffffffffull ? 0xffffffffu : (unsigned)eps;
    const bool wide = eps > 0xffffffffull;
#pragma unroll 4
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull w = FAST ? dbf_px_word_fast(xsh, sh0, n, eps32, wide, m, t0 + W * 64, lane)
                           : dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 + W * 64, lane);
        if (lane == 0) pm[1 + W] = w;
    }
    if (wave == 0) {
        const ull w = FAST ? dbf_px_word_fast(xsh, sh0, n, eps32, wide, m, t0 - 64, lane)
                           : dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 - 64, lane);
        if (lane == 0) pm[0] = w;
    }
    __syncthreads();
    if (wave == 0) {   // lane = word
        const ull cur = pm[1 + lane], prev = pm[lane];
        PM[1 + (size_t)tile * DBF_WORDS + lane] = cur;   // PM[0] = 0 (before the array)
        unsigned c = dbf_popc(dbf_starts_x(cur, prev));
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0) agg[tile] = mono_pack(0, c);
    }
}

__global__ __launch_bounds__(DBF_THREADS) void dbm_x_labels(const ull *__restrict__ PM, const ull *__restrict__ pre, int n,
                                                            const int *__restrict__ boff, int nb, int m, int *__restrict__ xlab,
                                                            unsigned *__restrict__ runbase, int *__restrict__ seg0, int *__restrict__ seg1) {
    __shared__ ull stS[DBF_WORDS], fS[DBF_WORDS], hdS[DBF_WORDS], tlS[DBF_WORDS];
    __shared__ unsigned baseS[DBF_WORDS];
    __shared__ unsigned shh[DBF_WAVES], shv[DBF_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int t0 = tile * DBF_TILE;
    const unsigned tpre = tile_prefix(pre, tile, gridDim.x, shh, shv);
    if (wave == 0) {   // lane = word: all the bit algebra of the tile in one go
        const size_t gw = (size_t)tile * DBF_WORDS + lane;   // PM index of the word BEFORE mine (PM is shifted by one)
        const ull prev = PM[gw], cur = PM[gw + 1], next = PM[gw + 2];   // the buffer holds whole tiles + 2, zero past the data
        const int g0 = t0 + lane * 64;
        const ull valid = g0 >= n ? 0ull : (n - g0 >= 64 ? ~0ull : dbf_lt(n - g0));
        const ull st = dbf_starts_x(cur, prev);
        const ull f = dbf_smear(cur, prev, m) & valid;            // label != -1: some p in [i-m+1, i]
        const bool fprev = g0 > 0 && (m >= 64 ? prev != 0 : (prev >> (64 - m)) != 0);           // found(g0-1)
        const bool fnext = g0 + 64 < n && ((next & 1ull) || (cur >> (65 - m)) != 0);            // found(g0+64)
        const bool snext = (next & 1ull) && !(cur >> 63);                                       // start(g0+64)
        const ull same_as_prev = ((f << 1) | (ull)fprev) & ~st;   // point i-1 carries the same run id
        const ull fn = (f >> 1) | ((ull)fnext << 63);
        const ull sn = (st >> 1) | ((ull)snext << 63);
        unsigned c = dbf_popc(st), enter, th, tv;
        wave_seg_scan(0, c, tpre, enter, th, tv);
        stS[lane] = st;
        fS[lane] = f;
        hdS[lane] = f & ~same_as_prev;
        tlS[lane] = f & ~(fn & ~sn);
        baseS[lane] = enter;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        if (g0 >= n) break;
        const int i = g0 + lane;
        const ull st = dbf_uni(stS[W]), f = dbf_uni(fS[W]), hd = dbf_uni(hdS[W]), tl = dbf_uni(tlS[W]);
        const unsigned run = baseS[W] + dbf_cnt_le(st, lane);      // inclusive count of run starts at i
        if (i < n) xlab[i] = ((f >> lane) & 1ull) ? (int)run - 1 : -1;
        if ((hd >> lane) & 1ull) seg0[run - 1] = i;
        if ((tl >> lane) & 1ull) seg1[run - 1] = i + 1;
        if (nb == 1) {
            if (i == n - 1) runbase[1] = run;
        } else if (i < n) {   // runs before every bucket (ids restart per bucket, tiddit_cluster.pyx:140-154)
            const int b = db_bucket(boff, nb, i);
            if (i + 1 == boff[b + 1])
                for (int bb = b + 1; bb <= nb && boff[bb] == i + 1; bb++) runbase[bb] = run;
        }
    }
}

#define DBM_YSH4 ((DBF_TILE + 68 + DBF_M_MAX) / 4)   // chunks staged for [t0-68, t0+TILE+m)

// py / head / bucket-start masks + per-tile (has, count) summary of sub-run starts since the last head
__global__ __launch_bounds__(DBF_THREADS) void dbm_y_masks(const int *__restrict__ xlab, const unsigned *__restrict__ ys, int n,
                                                           const int *__restrict__ boff, int nb, ull eps, int m, ull *__restrict__ PY,
                                                           ull *__restrict__ HM, ull *__restrict__ BM, ull *__restrict__ agg) {
    __shared__ __attribute__((aligned(16))) int lsh[DBM_YSH4 * 4];
    __shared__ __attribute__((aligned(16))) unsigned ysh[DBM_YSH4 * 4];
    __shared__ ull pm[DBF_WORDS + 1], hm[DBF_WORDS], bm[DBF_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int t0 = tile * DBF_TILE;
    const int sh0 = t0 - 68;
    {   // all loads first (independent, one round trip), then the LDS stores
        constexpr int NCH = (DBM_YSH4 + DBF_THREADS - 1) / DBF_THREADS;
        int4 lv[NCH];
        uint4 yv[NCH];
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int c = tid + k * DBF_THREADS;
            const int g = sh0 + 4 * c;
            if (c < DBM_YSH4 && g >= 0 && g + 4 <= n) {
                lv[k] = *reinterpret_cast<const int4 *>(xlab + g);
                yv[k] = *reinterpret_cast<const uint4 *>(ys + g);
            } else if (c < DBM_YSH4) {
                int l[4];
                unsigned y[4];
                for (int q = 0; q < 4; q++) {
                    const bool in = g + q >= 0 && g + q < n;
                    l[q] = in ? xlab[g + q] : -2;       // -2: outside the array (never equals a label)
                    y[q] = in ? ys[g + q] : 0u;
                }
                lv[k] = make_int4(l[0], l[1], l[2], l[3]);
                yv[k] = make_uint4(y[0], y[1], y[2], y[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int c = tid + k * DBF_THREADS;
            if (c < DBM_YSH4) {
                *reinterpret_cast<int4 *>(lsh + 4 * c) = lv[k];
                *reinterpret_cast<uint4 *>(ysh + 4 * c) = yv[k];
            }
        }
    }
    __syncthreads();
    const unsigned eps32 = eps > 0xffffffffull ? 0xffffffffu : (unsigned)eps;
    const bool wide = eps > 0xffffffffull;
    auto py_word = [&](int g0) -> ull {   // branch free: the staged range always covers o and o+m-1
        const int i = g0 + lane;
        const int o = i - sh0;
        const int l = lsh[o], l2 = lsh[o + m - 1];
        const unsigned d = ysh[o + m - 1] - ysh[o];
        // next = y[i+1:i+m] must lie inside the same x-cluster (clusters are contiguous); sorted => max is the last
        return __ballot(i >= 0 && i + m - 1 < n && l >= 0 && l2 == l && (wide || d < eps32));
    };
#pragma unroll 4
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        const int i = g0 + lane;
        const ull w = py_word(g0);
        bool head = false, bstart = false;
        if (i < n) {
            head = i == 0 || lsh[i - 1 - sh0] != lsh[i - sh0];
            bstart = nb == 1 ? i == 0 : i == boff[db_bucket(boff, nb, i)];
        }
        const ull h = __ballot(head), b = __ballot(bstart);
        if (lane == 0) {
            pm[1 + W] = w;
            hm[W] = h;
            bm[W] = b;
        }
    }
    if (wave == 0) {
        const ull w = py_word(t0 - 64);
        if (lane == 0) pm[0] = w;
    }
    __syncthreads();
    if (wave == 0) {   // lane = word
        const ull cur = pm[1 + lane], prev = pm[lane], h = hm[lane];
        const size_t gw = (size_t)tile * DBF_WORDS + lane;
        PY[1 + gw] = cur;
        HM[gw] = h;
        BM[gw] = bm[lane];
        const ull st = dbf_starts_y(cur, prev, h);
        unsigned enter, th, tv;
        wave_seg_scan(h != 0, (unsigned)dbf_popc(h ? dbf_from_last(st, h) : st), 0, enter, th, tv);
        if (lane == 0) agg[tile] = mono_pack(th, tv);
    }
}

// extra-start / sub-run-1 / labelled masks + per-tile (has, count) summary of extra starts since the bucket start
__global__ __launch_bounds__(DBF_THREADS) void dbm_y_mid(const ull *__restrict__ PY, const ull *__restrict__ HM, const ull *__restrict__ BM,
                                                         const ull *__restrict__ pre1, int n, int m, ull *__restrict__ EM,
                                                         ull *__restrict__ S1M, ull *__restrict__ FM, ull *__restrict__ agg2) {
    __shared__ ull stS[DBF_WORDS], fS[DBF_WORDS], hS[DBF_WORDS], emS[DBF_WORDS], s1S[DBF_WORDS];
    __shared__ unsigned cinS[DBF_WORDS];
    __shared__ unsigned shh[DBF_WAVES], shv[DBF_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int t0 = tile * DBF_TILE;
    const size_t gwl = (size_t)tile * DBF_WORDS + lane;
    const unsigned tpre = tile_prefix(pre1, tile, gridDim.x, shh, shv);
    if (wave == 0) {   // lane = word
        const ull prev = PY[gwl], cur = PY[gwl + 1], h = HM[gwl];
        const int g0 = t0 + lane * 64;
        const ull valid = g0 >= n ? 0ull : (n - g0 >= 64 ? ~0ull : dbf_lt(n - g0));
        const ull st = dbf_starts_y(cur, prev, h);
        unsigned enter, th, tv;
        wave_seg_scan(h != 0, (unsigned)dbf_popc(h ? dbf_from_last(st, h) : st), tpre, enter, th, tv);
        stS[lane] = st;
        fS[lane] = dbf_smear(cur, prev, m) & valid;
        hS[lane] = h;
        cinS[lane] = enter;      // starts of the open x-cluster before this word
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull st = dbf_uni(stS[W]), f = dbf_uni(fS[W]), h = dbf_uni(hS[W]);
        // starts of my cluster at positions <= lane: all starts <= lane minus those before my cluster's head
        const ull hle = h & dbf_le(lane);
        const unsigned c_le = dbf_cnt_le(st, lane);
        const unsigned ss = hle ? c_le - (unsigned)dbf_popc(st & ((1ull << (63 - __clzll((long long)hle))) - 1ull)) : cinS[W] + c_le;
        const ull e = __ballot(((st >> lane) & 1ull) && ss >= 2);      // a start that is not the first of its cluster
        const ull s1 = __ballot(((f >> lane) & 1ull) && ss == 1);      // labelled and in sub-run 1
        if (lane == 0) {
            emS[W] = e;
            s1S[W] = s1;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const ull e = emS[lane], b = BM[gwl];
        EM[gwl] = e;
        S1M[gwl] = s1S[lane];
        FM[gwl] = fS[lane];
        unsigned enter, th, tv;
        wave_seg_scan(b != 0, (unsigned)dbf_popc(b ? dbf_from_last(e, b) : e), 0, enter, th, tv);
        if (lane == 0) agg2[tile] = mono_pack(th, tv);
    }
}

__global__ __launch_bounds__(DBF_THREADS) void dbm_y_final(const int *__restrict__ xlab, const unsigned *__restrict__ ord,
                                                           const ull *__restrict__ BM, const ull *__restrict__ EM,
                                                           const ull *__restrict__ S1M, const ull *__restrict__ FM,
                                                           const ull *__restrict__ pre2, int n, const int *__restrict__ boff, int nb,
                                                           const unsigned *__restrict__ runbase, double *__restrict__ labels,
                                                           long long *__restrict__ last_id) {
    __shared__ ull bS[DBF_WORDS], eS[DBF_WORDS], s1S[DBF_WORDS], fS[DBF_WORDS];
    __shared__ unsigned einS[DBF_WORDS];
    __shared__ unsigned shh[DBF_WAVES], shv[DBF_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int t0 = tile * DBF_TILE;
    const unsigned tpre = tile_prefix(pre2, tile, gridDim.x, shh, shv);
    // this thread's labels / destinations: independent coalesced loads, issued before the mask work
    int lv[DBF_STEPS];
    unsigned ov[DBF_STEPS];
#pragma unroll
    for (int s = 0; s < DBF_STEPS; s++) {
        const int i = t0 + (wave * DBF_STEPS + s) * 64 + lane;
        lv[s] = i < n ? xlab[i] : -1;
        ov[s] = i < n ? ord[i] : 0u;
    }
    if (wave == 0) {   // lane = word
        const size_t gwl = (size_t)tile * DBF_WORDS + lane;
        const ull b = BM[gwl], e = EM[gwl];
        unsigned enter, th, tv;
        wave_seg_scan(b != 0, (unsigned)dbf_popc(b ? dbf_from_last(e, b) : e), tpre, enter, th, tv);
        bS[lane] = b;
        eS[lane] = e;
        s1S[lane] = S1M[gwl];
        fS[lane] = FM[gwl];
        einS[lane] = enter;     // extra starts of the open bucket before this word
    }
    __syncthreads();
    const unsigned rb1 = runbase[0];                       // single-bucket constants (uniform loads, once)
    const long long Rb1 = (long long)(runbase[1] - rb1);
#pragma unroll
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        if (g0 >= n) break;
        const int i = g0 + lane;
        const ull e = dbf_uni(eS[W]), b = dbf_uni(bS[W]), s1 = dbf_uni(s1S[W]), f = dbf_uni(fS[W]);
        const ull ble = b & dbf_le(lane);
        const unsigned e_le = dbf_cnt_le(e, lane);
        const unsigned E = ble ? e_le - (unsigned)dbf_popc(e & ((1ull << (63 - __clzll((long long)ble))) - 1ull)) : einS[W] + e_le;
        if (i < n) {
            const int l = lv[s];
            const int bk = db_bucket(boff, nb, i);
            const unsigned rb = nb == 1 ? rb1 : runbase[bk];
            const long long Rb = nb == 1 ? Rb1 : (long long)(runbase[bk + 1] - rb);
            double lab = -1.0;
            // sub-run 1 keeps the x id; the k-th extra start of the bucket owns id (R-1)+k   (DBSCAN.py:112-122)
            if ((f >> lane) & 1ull) lab = ((s1 >> lane) & 1ull) ? (double)((unsigned)l - rb) : (double)(Rb - 1 + (long long)E);
            labels[l >= 0 ? ov[s] : (unsigned)i] = lab;
            if (last_id && i + 1 == boff[bk + 1]) last_id[bk] = Rb - 1 + (long long)E;
        }
    }
}

