// Fused single-pass ("chained scan", decoupled look-back) kernels of the clustering path.
// Included by tdt_dbscan.hip.  Used when m <= DBF_M_MAX; the multi-kernel path stays as the general one.
//
// Every kernel walks tiles of 4096 points (tile index = an atomic ticket, so a tile's predecessors have
// always started).  All per-point predicates are booleans, so the tile-local scans are done on 64-bit
// ballot masks: lane = point, one coalesced load per array per 64 points, prefix counts are
// popcounts of masked ballots, run heads/tails are shifts of those masks — no per-thread item arrays,
// no shuffles, no LDS bank conflicts.  A tile publishes its aggregate, looks back over the predecessors'
// 8-byte status words until it meets an inclusive prefix (or a reset) and then finishes its own points.  A status word carries flag + payload in ONE naturally aligned 8-byte store/load
// (relaxed, agent scope: the load bypasses the per-CU L1), so no fences are needed; spins are bounded
// and report through an error word instead of hanging.
#pragma once

#define DBF_M_MAX 64
#ifndef DBF_THREADS
#define DBF_THREADS 256
#endif
#define DBF_WAVES (DBF_THREADS / 64)
#define DBF_STEPS (64 / DBF_WAVES)          // 64-element words per wave
#define DBF_WORDS (DBF_WAVES * DBF_STEPS)   // 64 words per tile
#define DBF_TILE (DBF_WORDS * 64)           // 4096 points
#define DBF_SPIN_LIMIT (1u << 24)

#define CS_AGG 1ull
#define CS_PREFIX 2ull

typedef unsigned long long ull;
#ifdef DBF_DEBUG_SPINS
__device__ unsigned long long dbg_ts[2][8192][4];
#define DBG_TS(k, tile, slot) do { if ((threadIdx.x & 63) == 0 && (tile) < 8192) dbg_ts[k][tile][slot] = wall_clock64(); } while (0)
#else
#define DBG_TS(k, tile, slot)
#endif

__device__ __forceinline__ ull cs_pack(ull flag, unsigned has, ull val) { return (flag << 62) | ((ull)(has & 1u) << 61) | (val & ((1ull << 61) - 1)); }
__device__ __forceinline__ void cs_store(ull *p, ull w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ ull cs_load(ull *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Segmented-sum monoid: state = (has_reset, value since the last reset).  combine(a then b) = b.has ? b : (a.has, a.val + b.val)
//
// Called by the 64 lanes of wave 0.  Publishes this tile's aggregate, looks back, publishes the inclusive
// prefix and returns the exclusive prefix (state after all earlier tiles).
__device__ __forceinline__ void cs_lookback(ull *status, int tile, unsigned agg_has, ull agg_val, unsigned &pre_has, ull &pre_val,
                                            int *err) {
    const int lane = threadIdx.x & 63;
    pre_has = 0;
    pre_val = 0;
#ifdef DBF_DEBUG_SPINS
    unsigned dbg_spins = 0, dbg_hops = 0;
    unsigned *dbg = (unsigned *)err + 4;
#endif
#ifdef DBF_EXP_NOLOOKBACK
    return;
#endif
    if (tile > 0) {
        if (lane == 0) cs_store(&status[tile], cs_pack(CS_AGG, agg_has, agg_val));
        int base = tile - 1;
        unsigned spins = 0;
#ifdef DBF_DEBUG_SPINS
        dbg_spins = 0; dbg_hops = 0;
#endif
        // 256 predecessors per hop (4 status words per lane, nearest first): the prefix front has to cross
        // all tiles in flight, so the number of ~1 us hops is what bounds the pass
        while (true) {
            ull w[4];
            bool inv[4], trm[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int idx = base - (lane * 4 + r);
                w[r] = idx >= 0 ? cs_load(&status[idx]) : cs_pack(CS_PREFIX, 0, 0);  // before tile 0: identity prefix
                inv[r] = (w[r] >> 62) == 0;
                trm[r] = (w[r] >> 62) == CS_PREFIX || ((w[r] >> 61) & 1ull);
            }
            // per lane: walk its 4 words nearest-first up to its first terminator
            ull lv = 0;
            bool linv = false, lterm = false;
            unsigned lhas = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!lterm) {
                    linv = linv || inv[r];
                    lv += w[r] & ((1ull << 61) - 1);
                    if (trm[r]) {
                        lterm = true;
                        lhas = (unsigned)((w[r] >> 61) & 1ull);
                    }
                }
            }
            const unsigned long long invalid = __ballot(linv);
            const unsigned long long term = __ballot(lterm);
            const int L = term ? __ffsll((long long)term) - 1 : 64;   // first lane holding a terminator
            const unsigned long long need = L >= 63 ? ~0ull : ((2ull << L) - 1ull);
            if (invalid & need) {
                // a nearer tile has not published yet.  Do NOT re-read the whole window in a loop (a thousand
                // waves doing that saturate the few L2 channels holding the status array and delay the very
                // stores they wait for): one lane polls the single nearest missing word, with a sleep.
                const int F = __ffsll((long long)(invalid & need)) - 1;     // nearest lane with a missing word
                int r0 = 0;
                if (lane == F) {
#pragma unroll
                    for (int r = 3; r >= 0; r--)
                        if (inv[r]) r0 = r;
                    ull *wp = &status[base - (lane * 4 + r0)];
                    while ((cs_load(wp) >> 62) == 0) {
#ifdef DBF_DEBUG_SPINS
                        dbg_spins++;
#endif
                        if (++spins > DBF_SPIN_LIMIT) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                spins = __shfl(spins, F);
                if (spins > DBF_SPIN_LIMIT) {
                    if (lane == 0) atomicOr(err, 1);
                    break;
                }
                continue;
            }
#ifdef DBF_DEBUG_SPINS
            dbg_hops++;
#endif
            ull v = (lane <= L) ? lv : 0ull;
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
            pre_val += v;
            if (term) {
                pre_has = __shfl(lhas, L);
                break;
            }
            base -= 256;
        }
    }
#ifdef DBF_DEBUG_SPINS
    if (lane == 0 && tile > 0) {
        dbg_spins = 0;
        for (int l = 0; l < 64; l++) dbg_spins += 0;
        atomicAdd(&dbg[0], dbg_spins);
        atomicMax(&dbg[1], dbg_spins);
        atomicAdd(&dbg[2], dbg_hops);
        atomicAdd(&dbg[3], 1u);
    }
#endif
    if (lane == 0) {
        const unsigned ih = agg_has | pre_has;
        const ull iv = agg_has ? agg_val : pre_val + agg_val;
        cs_store(&status[tile], cs_pack(CS_PREFIX, ih, iv));
    }
}

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

// p word: bit = p[g0 + lane]  (DBSCAN.py:41-51), 0 for points outside [0, n); x comes from the LDS stage
__device__ __forceinline__ ull dbf_px_word(const unsigned *xsh, int sh0, int n, const int *__restrict__ boff, int nb, ull eps, int m,
                                           int g0, int lane) {
    const int i = g0 + lane;
    bool p = false;
    if (i >= 0 && i < n) {
        const int bend = nb == 1 ? n : boff[db_bucket(boff, nb, i) + 1];
        if (i + m <= bend) {                          // `for i in range(0, len(data)-m+1)` (:39)
            const int hi = min(i + m, bend - 1);      // data[i+1:i+m+1] truncates at the array end (:43)
            const unsigned xi = xsh[i - sh0];
            unsigned maxd = 0;
            for (int q = i + 1; q <= hi; q++) maxd = max(maxd, db_absdiff(xsh[q - sh0], xi));
            p = (ull)maxd < eps;
        }
    }
    return __ballot(p);
}

__global__ __launch_bounds__(DBF_THREADS) void dbf_x(const unsigned *__restrict__ x, int n, const int *__restrict__ boff, int nb,
                                                     ull eps, int m, ull *status, DbfCtl *ctl, int *err, int *__restrict__ xlab,
                                                     unsigned *__restrict__ runbase, int *__restrict__ seg0, int *__restrict__ seg1) {
    __shared__ unsigned xsh[DBF_XSH];
    __shared__ ull pm[DBF_WORDS + 2];   // p masks: [0] = the word before the tile, [1..64], [65] = the word after
    __shared__ unsigned wcnt[DBF_WAVES];
    __shared__ int s_tile;
    __shared__ ull s_pre;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (int)atomicAdd(&ctl->ticket[0], 1u);
    __syncthreads();
    const int tile = s_tile;
    const int t0 = tile * DBF_TILE;
    if (t0 >= n) return;
    if (tid == 0) DBG_TS(0, tile, 0);
    // stage the tile (+ halos) with independent coalesced loads: one memory round trip instead of one per word
    const int sh0 = t0 - 64;
#pragma unroll
    for (int k = 0; k < (DBF_XSH + DBF_THREADS - 1) / DBF_THREADS; k++) {
        const int o = tid + k * DBF_THREADS;
        const int g = sh0 + o;
        if (o < DBF_XSH) xsh[o] = (g >= 0 && g < n) ? x[g] : 0u;
    }
    __syncthreads();
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull w = dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 + W * 64, lane);
        if (lane == 0) pm[1 + W] = w;
    }
    if (wave == 0) {
        const ull w = dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 - 64, lane);
        if (lane == 0) pm[0] = w;
    }
    if (wave == DBF_WAVES - 1) {
        const ull w = dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 + DBF_TILE, lane);
        if (lane == 0) pm[DBF_WORDS + 1] = w;
    }
    __syncthreads();
    // run starts (the `cluster` boolean going False -> True, :52-62) of this wave's 1024 points
    unsigned cnt = 0;
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull cur = pm[1 + W], prev = pm[W];
        cnt += dbf_popc(cur & ~((cur << 1) | (prev >> 63)));
    }
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    if (wave == 0) {
        unsigned ph;
        ull pv;
        DBG_TS(0, tile, 1);
        cs_lookback(status, tile, 0, (ull)wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3], ph, pv, err);
        DBG_TS(0, tile, 2);
        if (lane == 0) s_pre = pv;
    }
    __syncthreads();
    unsigned base = (unsigned)s_pre;     // run starts before this wave's first point
    for (int w = 0; w < wave; w++) base += wcnt[w];
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        if (g0 >= n) break;
        const int i = g0 + lane;
        const ull cur = pm[1 + W], prev = pm[W], next = pm[2 + W];
        const ull valid = n - g0 >= 64 ? ~0ull : dbf_lt(n - g0);
        const ull st = cur & ~((cur << 1) | (prev >> 63));
        const ull f = dbf_smear(cur, prev, m) & valid;     // label != -1: some p in [i-m+1, i]
        const unsigned run = base + dbf_popc(st & dbf_le(lane));      // inclusive count of run starts at i
        const bool found = (f >> lane) & 1ull;
        if (i < n) xlab[i] = found ? (int)run - 1 : -1;
        // neighbours: the point before the word and the point after it
        const bool fprev = m >= 64 ? prev != 0 : (prev >> (64 - m)) != 0;                    // found(g0-1), g0 > 0
        const bool nvalid = g0 + 64 < n;
        const bool fnext = nvalid && ((next & 1ull) || (cur >> (65 - m)) != 0);              // found(g0+64)
        const bool snext = (next & 1ull) && !(cur >> 63);                                    // start(g0+64)
        const ull same_as_prev = ((f << 1) | (ull)(fprev && g0 > 0)) & ~st;      // point i-1 carries the same run id
        const ull fn = (f >> 1) | ((ull)fnext << 63);
        const ull sn = (st >> 1) | ((ull)snext << 63);
        const ull headm = f & ~same_as_prev;
        const ull tailm = f & ~(fn & ~sn);
        if ((headm >> lane) & 1ull) seg0[run - 1] = i;
        if ((tailm >> lane) & 1ull) seg1[run - 1] = i + 1;
        // runs before every bucket (ids restart per bucket, tiddit_cluster.pyx:140-154)
        if (nb == 1) {
            if (i == n - 1) runbase[1] = run;
        } else if (i < n) {
            const int b = db_bucket(boff, nb, i);
            if (i + 1 == boff[b + 1])
                for (int bb = b + 1; bb <= nb && boff[bb] == i + 1; bb++) runbase[bb] = run;
        }
        base += dbf_popc(st);
    }
    if (wave == 0) DBG_TS(0, tile, 3);
}

// ------------------------------------------------------------------------------------------ y pass
#define DBF_YSH (DBF_TILE + 65 + DBF_M_MAX)
// window test on the sorted y (DBSCAN.py:90-99), sub-run starts (:101-110), relabel (:112-122), scatter.
// Two chained scans in one kernel: (1) sub-run starts counted from the head of every x-cluster,
// (2) "extra" sub-run starts (every start that is not the first of its cluster) counted from the start
// of every bucket — the k-th extra start of a bucket owns the id (R-1)+k.
__global__ __launch_bounds__(DBF_THREADS) void dbf_y(const int *__restrict__ xlab, const unsigned *__restrict__ ys,
                                                     const unsigned *__restrict__ ord, int n, const int *__restrict__ boff, int nb,
                                                     const unsigned *__restrict__ runbase, ull eps, int m, ull *status1, ull *status2,
                                                     DbfCtl *ctl, int *err, double *__restrict__ labels, long long *__restrict__ last_id) {
    __shared__ int lsh[DBF_YSH];        // xlab staged for [t0-65, t0+TILE+m)
    __shared__ unsigned ysh[DBF_YSH];   // sorted y, same range
    __shared__ ull pm[DBF_WORDS + 1];   // py masks, [0] = the word before the tile
    __shared__ ull hm[DBF_WORDS];       // x-cluster heads (label differs from the previous point)
    __shared__ ull bm[DBF_WORDS];       // bucket starts
    __shared__ ull em[DBF_WORDS];       // extra sub-run starts
    __shared__ ull s1m[DBF_WORDS];      // point belongs to sub-run 1 of its cluster
    __shared__ ull fm[DBF_WORDS];       // point is labelled (some py in [i-m+1, i])
    __shared__ unsigned whas[DBF_WAVES], wval[DBF_WAVES];
    __shared__ int s_tile;
    __shared__ ull s_pre;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (int)atomicAdd(&ctl->ticket[1], 1u);
    __syncthreads();
    const int tile = s_tile;
    const int t0 = tile * DBF_TILE;
    if (t0 >= n) return;

    // stage labels and sorted y of the tile (+ halos) with independent coalesced loads
    const int sh0 = t0 - 65;            // labels are needed from t0-64-1 (head test of the halo word)
#pragma unroll
    for (int k = 0; k < (DBF_YSH + DBF_THREADS - 1) / DBF_THREADS; k++) {
        const int o = tid + k * DBF_THREADS;
        const int g = sh0 + o;
        if (o < DBF_YSH) {
            const bool in = g >= 0 && g < n;
            lsh[o] = in ? xlab[g] : -2;     // -2: outside the array (never equals a label)
            ysh[o] = in ? ys[g] : 0u;
        }
    }
    __syncthreads();
    auto py_word = [&](int g0) -> ull {
        const int i = g0 + lane;
        bool p = false;
        if (i >= 0 && i + m - 1 < n) {
            const int l = lsh[i - sh0];
            // next = y[i+1:i+m] must lie inside the same x-cluster (clusters are contiguous); sorted => max is the last
            if (l >= 0 && lsh[i + m - 1 - sh0] == l) p = (ull)(ysh[i + m - 1 - sh0] - ysh[i - sh0]) < eps;
        }
        return __ballot(p);
    };
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
    // ---- scan 1: sub-run starts since the head of the x-cluster
    unsigned has = 0, val = 0;
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull cur = pm[1 + W], prev = pm[W], h = hm[W];
        const ull st = cur & (h | ~((cur << 1) | (prev >> 63)));   // sy = py && (head || !py[i-1])
        if (h) {
            has = 1;
            val = dbf_popc(st & (~0ull << (63 - __clzll((long long)h))));
        } else {
            val += dbf_popc(st);
        }
    }
    if (lane == 0) {
        whas[wave] = has;
        wval[wave] = val;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned th = 0;
        ull tv = 0;
        for (int w = 0; w < DBF_WAVES; w++) {
            if (whas[w]) { th = 1; tv = wval[w]; } else tv += wval[w];
        }
        unsigned ph;
        ull pv;
        cs_lookback(status1, tile, th, tv, ph, pv, err);
        if (lane == 0) s_pre = pv;
    }
    __syncthreads();
    unsigned cin = (unsigned)s_pre;   // starts of the open x-cluster before this wave's first point
    for (int w = 0; w < wave; w++) cin = whas[w] ? wval[w] : cin + wval[w];
    __syncthreads();                  // whas/wval are reused by scan 2
    unsigned has2 = 0, val2 = 0;
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        const ull cur = pm[1 + W], prev = pm[W], h = hm[W], b = bm[W];
        const ull st = cur & (h | ~((cur << 1) | (prev >> 63)));
        const ull valid = g0 >= n ? 0ull : (n - g0 >= 64 ? ~0ull : dbf_lt(n - g0));
        // count of starts of my cluster at positions <= lane
        const ull hle = h & dbf_le(lane);
        const unsigned ss = hle ? (unsigned)dbf_popc(st & dbf_le(lane) & (~0ull << (63 - __clzll((long long)hle))))
                                : cin + (unsigned)dbf_popc(st & dbf_le(lane));
        const bool is_st = (st >> lane) & 1ull;
        const ull f = dbf_smear(cur, prev, m) & valid;
        const bool found = (f >> lane) & 1ull;
        const ull e = __ballot(is_st && ss >= 2);
        const ull s1 = __ballot(found && ss == 1);
        if (lane == 0) {
            em[W] = e;
            s1m[W] = s1;
            fm[W] = f;
        }
        cin = h ? (unsigned)dbf_popc(st & (~0ull << (63 - __clzll((long long)h)))) : cin + (unsigned)dbf_popc(st);
        // ---- scan 2 summary: extra starts since the start of the bucket
        if (b) {
            has2 = 1;
            val2 = dbf_popc(e & (~0ull << (63 - __clzll((long long)b))));
        } else {
            val2 += dbf_popc(e);
        }
    }
    if (lane == 0) {
        whas[wave] = has2;
        wval[wave] = val2;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned th = 0;
        ull tv = 0;
        for (int w = 0; w < DBF_WAVES; w++) {
            if (whas[w]) { th = 1; tv = wval[w]; } else tv += wval[w];
        }
        unsigned ph;
        ull pv;
        cs_lookback(status2, tile, th, tv, ph, pv, err);
        if (lane == 0) s_pre = pv;
    }
    __syncthreads();
    unsigned ein = (unsigned)s_pre;   // extra starts of the open bucket before this wave's first point
    for (int w = 0; w < wave; w++) ein = whas[w] ? wval[w] : ein + wval[w];
    unsigned ordv[DBF_STEPS];         // destinations, fetched with independent loads ahead of the relabel loop
#pragma unroll
    for (int s = 0; s < DBF_STEPS; s++) {
        const int i = t0 + (wave * DBF_STEPS + s) * 64 + lane;
        ordv[s] = i < n ? ord[i] : 0u;
    }
#pragma unroll
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const int g0 = t0 + W * 64;
        if (g0 >= n) break;
        const int i = g0 + lane;
        const ull e = em[W], b = bm[W], s1 = s1m[W], f = fm[W];
        const ull ble = b & dbf_le(lane);
        const unsigned E = ble ? (unsigned)dbf_popc(e & dbf_le(lane) & (~0ull << (63 - __clzll((long long)ble))))
                               : ein + (unsigned)dbf_popc(e & dbf_le(lane));     // inclusive count of extra starts of my bucket
        if (i < n) {
            const int l = lsh[i - sh0];
            const int bk = db_bucket(boff, nb, i);
            const unsigned rb = runbase[bk];
            const long long Rb = (long long)(runbase[bk + 1] - rb);
            double lab = -1.0;
            if ((f >> lane) & 1ull) lab = ((s1 >> lane) & 1ull) ? (double)((unsigned)l - rb) : (double)(Rb - 1 + (long long)E);
            labels[l >= 0 ? ordv[s] : (unsigned)i] = lab;
            if (last_id && i + 1 == boff[bk + 1]) last_id[bk] = Rb - 1 + (long long)E;
        }
        ein = b ? (unsigned)dbf_popc(e & (~0ull << (63 - __clzll((long long)b)))) : ein + (unsigned)dbf_popc(e);
    }
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

// p masks of one tile (+ the count of run starts)  — x staged with 16-byte loads
__global__ __launch_bounds__(DBF_THREADS) void dbm_x_masks(const unsigned *__restrict__ x, int n, const int *__restrict__ boff, int nb,
                                                           ull eps, int m, ull *__restrict__ PM, ull *__restrict__ agg) {
    __shared__ __attribute__((aligned(16))) unsigned xsh[DBM_XSH4 * 4];
    __shared__ ull pm[DBF_WORDS + 1];   // [0] = last word of the previous tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
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
    for (int s = 0; s < DBF_STEPS; s++) {
        const int W = wave * DBF_STEPS + s;
        const ull w = dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 + W * 64, lane);
        if (lane == 0) pm[1 + W] = w;
    }
    if (wave == 0) {
        const ull w = dbf_px_word(xsh, sh0, n, boff, nb, eps, m, t0 - 64, lane);
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
    auto py_word = [&](int g0) -> ull {
        const int i = g0 + lane;
        bool p = false;
        if (i >= 0 && i + m - 1 < n) {
            const int l = lsh[i - sh0];
            // next = y[i+1:i+m] must lie inside the same x-cluster (clusters are contiguous); sorted => max is the last
            if (l >= 0 && lsh[i + m - 1 - sh0] == l) p = (ull)(ysh[i + m - 1 - sh0] - ysh[i - sh0]) < eps;
        }
        return __ballot(p);
    };
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
            const unsigned rb = runbase[bk];
            const long long Rb = (long long)(runbase[bk + 1] - rb);
            double lab = -1.0;
            // sub-run 1 keeps the x id; the k-th extra start of the bucket owns id (R-1)+k   (DBSCAN.py:112-122)
            if ((f >> lane) & 1ull) lab = ((s1 >> lane) & 1ull) ? (double)((unsigned)l - rb) : (double)(Rb - 1 + (long long)E);
            labels[l >= 0 ? ov[s] : (unsigned)i] = lab;
            if (last_id && i + 1 == boff[bk + 1]) last_id[bk] = Rb - 1 + (long long)E;
        }
    }
}

