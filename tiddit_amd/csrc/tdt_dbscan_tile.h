// Tile-resident clustering pass.  Included by tdt_dbscan.hip; used when m <= DBF_M_MAX and no x-cluster has more than
// DB_SMALL members (the rare other cases fall back to the multi-launch path in tdt_dbscan.hip).
//
// The closed form of DBSCAN.py (see the header of tdt_dbscan.hip) has exactly two global dependencies: the number of
// x-runs before a point (its x id) and, for the relabelling of sub-runs, the number of runs of the whole bucket plus the
// number of extra sub-runs before the point (DBSCAN.py:112-122).  Everything else is local to an x-cluster, and x-clusters
// are contiguous index ranges.  So ONE launch does all the local work with the tile in LDS:
//
//   dbt_tile    a workgroup stages DT_NW words (1536 points) of x and y, OWNS the x-clusters that start in all but its last two words
//               (a cluster may run on into the two halo words: it has at most DB_SMALL = 128 members) and computes for them
//               the window masks p (DBSCAN.py:41-51), runs / labelled masks (:52-62), the stable y order of every cluster
//               (:76-81), the y window masks (:90-99), sub-run starts and sub-run numbers (:101-110).  Every point gets a
//               2-byte CODE in a side array: -1, or (kind, owner-tile flag, tile-local index).
//               Per tile it leaves two counts: x-runs started, extra sub-runs.
//               (The first workgroup of the NEXT launch stores the word that tells the host whether a cluster was too large for
//               this path into pinned memory: no extra launch, no stream synchronisation, no per-workgroup fence.)
//   dbt_finish1 (one bucket) code -> float64 id: a workgroup per tile sums the count arrays itself (its two prefixes, the totals).
//   dbt_scan + dbt_finish (several buckets) one workgroup scans the counts and forms the per-bucket bases and last_id; then
//               code -> float64 id with ids restarting per bucket.
//
// All predicates are ballot masks (lane = point), the word-level algebra runs with lane = word, as in tdt_dbscan_fused.h.
#pragma once
#include <type_traits>

#ifndef DT_THREADS
#define DT_THREADS 256
#endif
#define DT_WAVES (DT_THREADS / 64)
#ifndef DT_NW
#define DT_NW 24                            // staged words per tile
#endif
#define DT_OW (DT_NW - 2)                   // owned words: clusters starting here are this tile's
#define DT_S (DT_NW * 64)                   // 1536 staged positions (DT_NW 24)
#define DT_T (DT_OW * 64)                   // 1408 owned positions
#define DT_WPW (DT_NW / DT_WAVES)           // words per wave
#define DT_XS (DT_S + 64 + DBF_M_MAX + 8)   // x staged for [t0-64, t0+S+m+...)
// a point's 2-byte CODE between the tile kernel and the finish kernel (its own array: 2 B/point written and read once, the
// float64 labels written once):
#define DT_C_MINUS1 0xffffu                 // label -1.0
#define DT_C_PREV 0x4000u                   // the position belongs to the NEXT tile's range: its owner is tile(position) - 1
#define DT_C_EXTRA 0x2000u                  // index counts extra sub-run starts (else: x-run index)
#define DT_C_LITERAL 0x1000u                // the id is the caller-supplied x label of the position
#define DT_C_INDEX 0x0fffu                  // tile-local index (x-runs / extra starts of a tile: at most DT_S / 2)
#define DT_MINUS1 0xbff0000000000000ull     // bits of -1.0
#define DT_GRP 64                           // tiles per group of the two-level count sums (one-bucket path)
#define DT_GRPMAX (0x7fffffff / (DT_GRP * DT_T) + 2)   // groups of tiles a call can have (n < 2^31)

#ifdef DT_PROF
// variant builds only (tools/dt_prof.sh): shader-clock cycles per phase of dbt_tile as seen by thread 0, per workgroup
#define DT_PROF_TILES 8192
__device__ unsigned dt_prof[DT_PROF_TILES * 16];
#define DT_MARK(k) do { if (ONE_BUCKET && !LABELS && tid == 0 && tile < DT_PROF_TILES) { const unsigned long long t_ = clock64(); dt_prof[tile * 16 + k] = (unsigned)(t_ - t_last); t_last = t_; } } while (0)
#else
#define DT_MARK(k) do { } while (0)
#endif

struct DtParams {
    const unsigned *x, *y;
    int n;
    const int *boff;
    int nb;
    unsigned eps32;
    int wide;                 // eps > 2^32-1: every 32-bit distance qualifies
    int m;
    unsigned short *code;     // n codes (DT_C_*)
    unsigned *aggR, *aggE;    // per tile: x-runs started in the owned words, extra sub-run starts of the owned clusters
    unsigned *brun, *bext;    // per bucket whose first point lies in the tile: the two counts in front of that point
    unsigned *flags;          // [0] != 0: some x-cluster is too large for this path
    unsigned *grp;            // one-bucket path: sums of aggR / aggE over groups of DT_GRP tiles ([2][DT_GRPMAX], zero on entry)
};

// the kernel that follows the tile kernel tells the host whether the pass stands (no extra launch, no stream synchronisation)
__device__ __forceinline__ void dt_signal_host(unsigned *flags, volatile unsigned *host, unsigned seq) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && host) {
        host[0] = flags[0];
        flags[0] = 0;                      // ready for the next call (nothing else touches it before the next tile kernel)
        flags[1] = 0;                      // ... and so is the tile counter of the persistent grid
        __threadfence_system();
        host[1] = seq;
    }
}

// bits [pos+1, pos+cnt] of the bit stream formed by words w0 (holding pos), w1, w2; 1 <= cnt <= 64
__device__ __forceinline__ ull dt_bits_after(ull w0, ull w1, ull w2, int bit, int cnt) {
    // stream position of bit+1 inside w0; shifts by 64 are avoided by splitting
    const int s = bit + 1;                                  // 1..64
    ull v = s < 64 ? (w0 >> s) | (w1 << (64 - s)) : w1;
    (void)w2;
    return cnt >= 64 ? v : v & ((1ull << cnt) - 1ull);
}

// lanes where a - b - cin borrows, i.e. a < b + cin (cin: one bit per lane); no flags register is touched
__device__ __forceinline__ ull dt_borrow(unsigned a, unsigned b, ull cin) {
    unsigned diff;
    ull bout;
    asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(diff), "=s"(bout) : "v"(a), "v"(b), "s"(cin));
    return bout;
}
// r + (the lane's bit of mask)
__device__ __forceinline__ unsigned dt_add_bit(unsigned r, ull mask) {
    unsigned out;
    ull cout;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(out), "=s"(cout) : "v"(r), "s"(mask));
    return out;
}

// LABELS: the x pass is not computed — P.x holds caller-supplied x labels (int32, -1 = unlabelled; every label one contiguous range),
// DBSCAN.y_coordinate_clustering's `clusters` argument (DBSCAN.py:66-74); sub-run 1 keeps the label, extra sub-runs are numbered
// from the caller's cluster_id by dbt_finish1.
template <bool ONE_BUCKET, bool XONLY, bool LABELS = false>
__device__ __forceinline__ void dbt_tile_body(const DtParams &P, const int tile) {
    __shared__ __attribute__((aligned(16))) unsigned xs[DT_XS];        // x, later the y values in sorted order
    __shared__ __attribute__((aligned(16))) unsigned yv[DT_S + 8];      // + 8: the rank loop's masked reads past the last cluster
    // run starts lie at least two positions apart (a start needs a non-p position in front of it), so the staged range holds at
    // most DT_S / 2 x-clusters; caller-supplied labels may start one at every position
    constexpr int NSEG = LABELS ? DT_S : DT_S / 2 + 2;
    __shared__ unsigned short ordl[DT_S], segA[NSEG], segE[NSEG];
    __shared__ ull PM[DT_NW + 2], ST[DT_NW], FM[DT_NW], TL[DT_NW], FY[DT_NW], BM[DT_NW + 3];
    ull *PY = PM;                        // the x window masks are dead once the run masks exist; so are the cluster tails (8 workgroups
    ull *EB = TL;                        // of 20 KB fit a CU's LDS, not 6 of 23 KB)
    ull *SY = BM;                        // the bucket-boundary stream is read by the x pass only
    __shared__ unsigned runBase[DT_NW + 1], extBase[DT_NW + 1];
    __shared__ unsigned s_owned, s_b0, s_b1;
    int tid_ = threadIdx.x;
#ifdef DT_PERSIST
    // (inlined into the tile loop: without this opaque copy the compiler hoists everything derived from the thread index out of the
    // loop and keeps it live across tiles — 64 VGPRs + 22 spilled to scratch against 42)
    asm volatile("" : "+v"(tid_));
#endif
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, m = P.m;
    const int t0 = tile * DT_T;          // the host takes this path only for n < 2^31 - 2^16: int arithmetic cannot overflow
    const int sh0 = t0 - 64;
    unsigned *ysrt = xs;
#ifdef DT_PROF
    unsigned long long t_last = clock64();
#endif

    // ---- stage x [t0-64, ...) and y [t0, t0+S) with 16-byte loads, zero outside the array
    {
        constexpr int NCX = (DT_XS / 4 + DT_THREADS - 1) / DT_THREADS;
        uint4 v[NCX];
#pragma unroll
        for (int k = 0; k < NCX; k++) {
            const int c = tid + k * DT_THREADS;
            const int g = sh0 + 4 * c;
            constexpr unsigned FILL = LABELS ? 0xffffffffu : 0u;      // outside the array: x 0 / label -1
            v[k] = make_uint4(FILL, FILL, FILL, FILL);
            if (c < DT_XS / 4) {
                if (g >= 0 && g + 4 <= n) v[k] = *reinterpret_cast<const uint4 *>(P.x + g);
                else {
                    v[k].x = (g >= 0 && g < n) ? P.x[g] : FILL;
                    v[k].y = (g + 1 >= 0 && g + 1 < n) ? P.x[g + 1] : FILL;
                    v[k].z = (g + 2 >= 0 && g + 2 < n) ? P.x[g + 2] : FILL;
                    v[k].w = (g + 3 >= 0 && g + 3 < n) ? P.x[g + 3] : FILL;
                }
            }
        }
        constexpr int NCY = (DT_S / 4 + DT_THREADS - 1) / DT_THREADS;
        uint4 w[NCY];
        if (!XONLY) {
#pragma unroll
            for (int k = 0; k < NCY; k++) {
                const int c = tid + k * DT_THREADS;
                const int g = t0 + 4 * c;
                w[k] = make_uint4(0, 0, 0, 0);
                if (c >= DT_S / 4) continue;
                if (g + 4 <= n) w[k] = *reinterpret_cast<const uint4 *>(P.y + g);
                else {
                    w[k].x = g < n ? P.y[g] : 0u;
                    w[k].y = g + 1 < n ? P.y[g + 1] : 0u;
                    w[k].z = g + 2 < n ? P.y[g + 2] : 0u;
                    w[k].w = g + 3 < n ? P.y[g + 3] : 0u;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NCX; k++) {
            const int c = tid + k * DT_THREADS;
            if (c < DT_XS / 4) *reinterpret_cast<uint4 *>(xs + 4 * c) = v[k];
        }
        if (!XONLY) {
#pragma unroll
            for (int k = 0; k < NCY; k++)
                if (tid + k * DT_THREADS < DT_S / 4) *reinterpret_cast<uint4 *>(yv + 4 * (tid + k * DT_THREADS)) = w[k];
        }
    }
    // bucket boundaries (first points of buckets, and n) inside [t0-64, t0+S+128) as a bit stream: BM word k covers
    // positions [t0-64+64k, ...)
    if (!ONE_BUCKET) {
        if (tid < DT_NW + 3) BM[tid] = 0;
        if (tid == 0) {
            int lo = 0, hi = P.nb;                 // first boundary index with boff[b] >= sh0
            if (P.boff[0] >= sh0) hi = 0;
            else {
                while (hi - lo > 1) {              // boff[lo] < sh0 <= boff[hi]  (boff[nb] = n; a tile exists only for t0 < n)
                    const int mid = (lo + hi) >> 1;
                    if (P.boff[mid] < sh0) lo = mid;
                    else hi = mid;
                }
            }
            int b1 = hi;
            while (b1 <= P.nb && P.boff[b1] < sh0 + 64 * (DT_NW + 3)) b1++;
            s_b0 = (unsigned)hi;
            s_b1 = (unsigned)b1;
        }
        __syncthreads();
        for (unsigned b = s_b0 + tid; b < s_b1; b += DT_THREADS) {
            const int off = P.boff[b] - sh0;
            atomicOr(&BM[off >> 6], 1ull << (off & 63));
        }
    }
    __syncthreads();
    DT_MARK(0);

    // ---- x pass: p words -1 .. NW-1   (PM[1 + W]; PM[0] = the word before the tile)
    auto p_word = [&](int W, auto m3) -> ull {                // m3: m == 3, the caller's usual l (tiddit_cluster.pyx: min_pts = 3)
        const int i = t0 + 64 * W + lane;                     // global index
        const int o = 64 * W + lane + 64;                     // xs index
        const unsigned xi = xs[o];
        // Only the window's LAST member can be missing where p can hold: i <= n-m admits i = n-m, whose window data[i+1:i+m+1] is
        // one short (:39,43); a bucket boundary at i+m does the same.  The first m-1 members are compared unconditionally (outside
        // the array the staging wrote zeros, and p is false there anyway).
        bool p, last;
        if (ONE_BUCKET) {
            last = i + m < n;
            p = i >= 0 && i + m <= n;
        } else {
            // boundaries at positions i+1 .. i+m: one among the first m-1 ends the bucket inside the window (no p); one at
            // i+m exactly truncates the window by one member
            const int k = (64 * W + lane + 64) >> 6, bit = (64 * W + lane + 64) & 63;
            const ull nb_bits = dt_bits_after(BM[k], BM[k + 1], 0, bit, m);
            last = !((nb_bits >> (m - 1)) & 1ull);
            p = i >= 0 && i < n && (nb_bits & ((1ull << (m - 1)) - 1ull)) == 0;
        }
        // max |x_j - x_i| over the window from the window's largest and smallest value (the input need not be sorted: :41-45 takes abs)
        unsigned hi = xi, lo = xi;
        if (decltype(m3)::value) {
            const unsigned v1 = xs[o + 1], v2 = xs[o + 2], r3 = xs[o + 3], v3 = last ? r3 : xi;      // (loads stay unconditional)
            hi = max(max(v1, v2), max(v3, xi));
            lo = min(min(v1, v2), min(v3, xi));
        } else {
            for (int q = 1; q < m; q++) {
                const unsigned v = xs[o + q];
                hi = max(hi, v);
                lo = min(lo, v);
            }
            const unsigned rl = xs[o + m], v = last ? rl : xi;
            hi = max(hi, v);
            lo = min(lo, v);
        }
        const unsigned maxd = max(hi - xi, xi - lo);
        return __ballot(p & (P.wide != 0 | maxd < P.eps32));       // (no short circuit: the words of a wave interleave)
    };
    if (LABELS) {
        // caller-supplied labels: "labelled" and "a cluster starts here" straight from them
#pragma unroll 4
        for (int s = 0; s < DT_WPW; s++) {
            const int W = wave * DT_WPW + s;
            const int o = 64 * W + lane + 64;
            const int l = (int)xs[o], lp = (int)xs[o - 1];
            const ull fw = __ballot(l >= 0), sw = __ballot(l >= 0 && l != lp);
            if (lane == 0) {
                FM[W] = fw;
                ST[W] = sw;
            }
        }
    } else {
        auto x_pass = [&](auto m3) {
            ull w[DT_WPW];
#pragma unroll
            for (int s = 0; s < DT_WPW; s++) w[s] = p_word(wave * DT_WPW + s, m3);
            if (lane == 0) {
#pragma unroll
                for (int s = 0; s < DT_WPW; s++) PM[1 + wave * DT_WPW + s] = w[s];
            }
            if (wave == 0) {
                const ull w1 = p_word(-1, m3);
                if (lane == 0) PM[0] = w1;
            }
        };
        if (m == 3) x_pass(std::true_type{});
        else x_pass(std::false_type{});
    }
    __syncthreads();
    DT_MARK(1);

    // ---- lane = word: run starts, labelled mask, cluster tails; counts of starts before every word
    if (wave == 0) {
        const int W = lane;
        const bool act = W < DT_NW;
        const ull cur = (act && !LABELS) ? PM[1 + W] : 0ull, prev = (act && !LABELS) ? PM[W] : 0ull;
        const int g0 = t0 + 64 * W;
        const ull valid = g0 >= n ? 0ull : (n - g0 >= 64 ? ~0ull : dbf_lt(n - g0));
        const ull st = LABELS ? (act ? ST[W] : 0ull) : cur & ~((cur << 1) | (prev >> 63));     // a run starts: p and not p before (:52-57)
        const ull f = LABELS ? (act ? FM[W] : 0ull) : dbf_smear(cur, prev, m) & valid;         // label != -1: some p in [i-m+1, i]  (:58-62)
        // f / st of the next word's first position (the last word's cluster is closed by force: a cluster reaching the end of
        // the staged range has more than DB_SMALL members, since it started in the owned words)
        ull fnx = __shfl_down(f, 1), snx = __shfl_down(st, 1);
        if (W == DT_NW - 1) { fnx = 0; snx = 0; }
        const ull fn = (f >> 1) | (fnx << 63), sn = (st >> 1) | (snx << 63);
        const ull tl = f & ~(fn & ~sn);                                    // last member: the next point is unlabelled or starts a run
        if (act) {
            ST[W] = st;
            FM[W] = f;
            TL[W] = tl;
        }
        unsigned c = (unsigned)dbf_popc(st), incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (act) runBase[W] = incl - c;
        const unsigned owned = __shfl(incl, DT_OW - 1);
        if (lane == 0) {
            s_owned = owned;
            P.aggR[tile] = owned;
            if (ONE_BUCKET && owned) atomicAdd(&P.grp[tile / DT_GRP], owned);
        }
    }
    __syncthreads();
    DT_MARK(2);
    const unsigned n_owned = s_owned;

    // ---- lane = point: cluster ids and extents.  cid = index (1-based) of the point's cluster among the runs started in the
    // staged range; 1..n_owned are this tile's.  Packed per point for the later steps: cid << 1 | labelled.
    unsigned info[DT_WPW];
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        const int q = 64 * W + lane;
        const ull st = dbf_uni(ST[W]), f = dbf_uni(FM[W]), tl = dbf_uni(TL[W]);
        const unsigned cid = runBase[W] + dbf_cnt_le(st, lane);
        const bool lab = (f >> lane) & 1ull;
        const bool mine = lab && cid >= 1 && cid <= n_owned;
        info[s] = mine ? (cid << 1 | 1u) : 0u;
        if (mine && ((st >> lane) & 1ull)) segA[cid - 1] = (unsigned short)q;
        if (mine && ((tl >> lane) & 1ull)) segE[cid - 1] = (unsigned short)(q + 1);
        if (XONLY) {
            if (mine && q == DT_S - 1) atomicOr(P.flags, 1u);     // the cluster may run past the staged range: not this path's case
            const int g = t0 + q;
            if (g < n) {
                if (mine) P.code[g] = (unsigned short)((q >= DT_T ? DT_C_PREV : 0u) | (cid - 1));
                else if (!lab && q < DT_T) P.code[g] = (unsigned short)DT_C_MINUS1;
            }
        }
    }
    if (XONLY) {
        if (!ONE_BUCKET) {
            for (unsigned b = s_b0 + tid; b < s_b1; b += DT_THREADS) {
                const int off = P.boff[b] - t0;
                if (off >= 0 && off < DT_T && P.boff[b] < n) {
                    const int W = off >> 6, bit = off & 63;
                    P.brun[b] = runBase[W] + (unsigned)dbf_popc(ST[W] & dbf_lt(bit));
                    P.bext[b] = 0;
                }
            }
        }
        if (tid == 0) P.aggE[tile] = 0;
        return;
    }
    __syncthreads();
    DT_MARK(3);

    // ---- stable y order inside every owned cluster (:76-81): rank = members sorting before the point
    unsigned ext[DT_WPW];      // a | e << 16 of the point's cluster (0: not a member of an owned small cluster)
    bool large = false;
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        ext[s] = 0;
        if (info[s]) {
            const unsigned cid = info[s] >> 1;
            const int a = segA[cid - 1], e = segE[cid - 1];
            if (e - a > DB_SMALL) large = true;
            else ext[s] = (unsigned)a | ((unsigned)e << 16);
        }
    }
    __syncthreads();           // every x read is done: xs becomes ysrt
    DT_MARK(4);
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        const int q = 64 * W + lane;
        if (ext[s]) {
            const int a = ext[s] & 0xffff, e = ext[s] >> 16;
            const unsigned yq = yv[q];
            int rank = 0;
            // 8 members per trip, branch-free: one address, eight loads at constant offsets (reads past the cluster's end stay inside
            // the workgroup's LDS and are masked).  Per member four vector instructions: two compares of the trip's distances with a
            // constant (masks in SGPRs), one subtract-with-borrow whose borrow-out IS the sort predicate, one add-with-carry
            // (the wave runs as long as its largest cluster; a cluster of the usual size is done in one trip)
            for (int j0 = a; j0 < e; j0 += 8) {
                const int dq = q - j0, de = e - j0;                 // member k sorts before q on a tie iff k < dq; it exists iff k < de
                unsigned v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = yv[j0 + k];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const ull tie = __ballot(k < dq), there = __ballot(k < de);
                    rank = (int)dt_add_bit((unsigned)rank, dt_borrow(v[k], yq, tie) & there);      // v < yq + tie
                }
            }
            ysrt[a + rank] = yq;
            ordl[a + rank] = (unsigned short)q;
        }
    }
    if (large) atomicOr(P.flags, 1u);
    __syncthreads();
    DT_MARK(5);

    // ---- y pass on the sorted values: window test with m-1 following members (:90-99)
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        const int q = 64 * W + lane;
        bool py = false;
        if (ext[s]) {
            const int e = ext[s] >> 16;
            if (q + m <= e) py = P.wide || (ysrt[q + m - 1] - ysrt[q] < P.eps32);
        }
        const ull w = __ballot(py);
        if (lane == 0) PY[1 + W] = w;
    }
    if (tid == 0) PY[0] = 0;
    __syncthreads();
    DT_MARK(6);
    if (wave == 0) {   // lane = word: sub-run starts (a cluster head always starts one), labelled mask, starts before every word
        const int W = lane;
        ull sy = 0;
        if (W < DT_NW) {
            const ull cur = PY[1 + W], prev = PY[W], h = ST[W];
            sy = cur & (h | ~((cur << 1) | (prev >> 63)));
            SY[W] = sy;
            FY[W] = dbf_smear(cur, prev, m);
        }
        const unsigned c = (unsigned)dbf_popc(sy);
        unsigned incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (W < DT_NW) extBase[W] = incl - c;      // (the array holds the extra-start counts later)
    }
    __syncthreads();
    DT_MARK(7);
    // sub-run number of a member = sub-run starts in [a, q] = starts up to q minus starts before the cluster's head, which the
    // head publishes; a start that is not the first of its cluster is an EXTRA (:112-122)
    unsigned short *cHead = segA;                  // the cluster extents live in registers by now
    unsigned sub[DT_WPW];
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        const int q = 64 * W + lane;
        const ull sy = dbf_uni(SY[W]);
        sub[s] = extBase[W] + dbf_cnt_le(sy, lane);
        if (ext[s] && (int)(ext[s] & 0xffff) == q) cHead[(info[s] >> 1) - 1] = (unsigned short)(sub[s] - (unsigned)((sy >> lane) & 1ull));
    }
    __syncthreads();
    DT_MARK(8);
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        bool extra = false;
        if (ext[s]) {
            sub[s] -= cHead[(info[s] >> 1) - 1];
            extra = ((SY[W] >> lane) & 1ull) && sub[s] >= 2;
        }
        const ull w = __ballot(extra);
        if (lane == 0) EB[W] = w;
    }
    __syncthreads();
    DT_MARK(9);
    if (wave == 0) {
        const unsigned c = lane < DT_NW ? (unsigned)dbf_popc(EB[lane]) : 0u;
        unsigned incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane < DT_NW) extBase[lane] = incl - c;
        if (lane == 63) {
            P.aggE[tile] = incl;
            if (ONE_BUCKET && incl) atomicAdd(&P.grp[DT_GRPMAX + tile / DT_GRP], incl);
        }
    }
    __syncthreads();
    DT_MARK(10);

    // ---- results: -1.0 or a code, written at the member's ORIGINAL position (labels come back in input order)
#pragma unroll
    for (int s = 0; s < DT_WPW; s++) {
        const int W = wave * DT_WPW + s;
        const int q = 64 * W + lane;
        if (ext[s]) {
            const unsigned dq = ordl[q];                       // the point sorted to position q
            const int g = t0 + (int)dq;
            unsigned code = DT_C_MINUS1;
            if ((FY[W] >> lane) & 1ull) {
                const unsigned prevf = dq >= DT_T ? DT_C_PREV : 0u;
                if (sub[s] == 1) code = LABELS ? DT_C_LITERAL : (prevf | ((info[s] >> 1) - 1u));   // sub-run 1 keeps the x id
                else code = prevf | DT_C_EXTRA | (extBase[W] + dbf_cnt_le(EB[W], lane));           // k-th extra start of the bucket
            }
            P.code[g] = (unsigned short)code;
        } else if (!info[s] && q < DT_T && t0 + q < n && !((FM[W] >> lane) & 1ull)) {
            P.code[t0 + q] = (unsigned short)DT_C_MINUS1;       // not in any x-cluster
        }
    }
    DT_MARK(11);
    if (!ONE_BUCKET) {
        for (unsigned b = s_b0 + tid; b < s_b1; b += DT_THREADS) {
            const int off = P.boff[b] - t0;
            if (off >= 0 && off < DT_T && P.boff[b] < n) {
                const int W = off >> 6, bit = off & 63;
                P.brun[b] = runBase[W] + (unsigned)dbf_popc(ST[W] & dbf_lt(bit));
                P.bext[b] = extBase[W] + (unsigned)dbf_popc(EB[W] & dbf_lt(bit));
            }
        }
    }
}

// The launch: one workgroup per tile.  5 M points are 3552 tiles on 2048 resident slots — 1.73 "rounds", the second one 27 % empty — so a
// PERSISTENT grid (as many workgroups as the chip holds, each taking tiles off one counter, P.flags[1]) was measured in round 4
// (-DDT_PERSIST, tools/ab_db.sh): 93-100 us between events against 65-67 us, with or without the register spills the tile loop first
// caused (64 VGPRs + 22 spilled; 57 and none with the opaque thread index below).  The tail round is cheaper than the loop: kept off.
template <bool ONE_BUCKET, bool XONLY, bool LABELS = false>
__global__ __launch_bounds__(DT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void dbt_tile(DtParams P) {
#ifndef DT_PERSIST
    dbt_tile_body<ONE_BUCKET, XONLY, LABELS>(P, (int)blockIdx.x);
#else
    __shared__ int s_tile;
    const int ntiles = (P.n + DT_T - 1) / DT_T;
    for (;;) {
        __syncthreads();                     // the previous tile's last LDS reads are done
        if (threadIdx.x == 0) s_tile = (int)atomicAdd(P.flags + 1, 1u);
        __syncthreads();
        const int tile = s_tile;
        if (tile >= ntiles) return;
        dbt_tile_body<ONE_BUCKET, XONLY, LABELS>(P, tile);
    }
#endif
}

// exclusive scans of the per-tile counts; bases of every bucket; last_id; the host's status word
__global__ __launch_bounds__(1024) void dbt_scan(unsigned *aggR, unsigned *aggE, int nt, const int *__restrict__ boff, int nb, int n,
                                                 const unsigned *__restrict__ brun, const unsigned *__restrict__ bext,
                                                 unsigned *__restrict__ runbase, unsigned *__restrict__ extbase, long long *__restrict__ last_id,
                                                 int xonly, unsigned *__restrict__ flags, volatile unsigned *host, unsigned seq) {
    __shared__ unsigned wr[16], we[16], cr, ce;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dt_signal_host(flags, host, seq);
    if (tid == 0) {
        cr = 0;
        ce = 0;
    }
    __syncthreads();
    for (int base = 0; base < nt; base += 1024) {
        const int i = base + tid;
        const unsigned r = i < nt ? aggR[i] : 0u, e = i < nt ? aggE[i] : 0u;
        unsigned sr = r, se = e;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned a = __shfl_up(sr, d), b = __shfl_up(se, d);
            if (lane >= d) {
                sr += a;
                se += b;
            }
        }
        if (lane == 63) {
            wr[wave] = sr;
            we[wave] = se;
        }
        __syncthreads();
        unsigned or_ = cr, oe = ce;
        for (int w = 0; w < wave; w++) {
            or_ += wr[w];
            oe += we[w];
        }
        if (i < nt) {
            aggR[i] = or_ + sr - r;
            aggE[i] = oe + se - e;
        }
        __syncthreads();
        if (tid == 1023) {
            cr = or_ + sr;
            ce = oe + se;
        }
        __syncthreads();
    }
    const unsigned totR = cr, totE = ce;
    for (int b = tid; b <= nb; b += 1024) {
        const int s = boff[b];
        unsigned rb = totR, eb = totE;
        if (s < n) {
            const int t = s / DT_T;
            rb = aggR[t] + brun[b];
            eb = aggE[t] + bext[b];
        }
        runbase[b] = rb;
        extbase[b] = eb;
    }
    __syncthreads();
    if (last_id)
        for (int b = tid; b < nb; b += 1024)
            last_id[b] = (long long)(runbase[b + 1] - runbase[b]) - 1 + (xonly ? 0ll : (long long)(extbase[b + 1] - extbase[b]));
}

__global__ __launch_bounds__(256) void dbt_finish(const unsigned short *__restrict__ code, double *__restrict__ lab, int n,
                                                  const unsigned *__restrict__ preR, const unsigned *__restrict__ preE,
                                                  const int *__restrict__ boff, int nb, const unsigned *__restrict__ runbase,
                                                  const unsigned *__restrict__ extbase) {
    // two pairs of points per thread, pair h of thread t at block*1024 + h*512 + 2t: a wave's store instruction writes one contiguous KB
    const bool al = (((size_t)lab) & 15) == 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const long long i0 = (long long)blockIdx.x * 1024 + h * 512 + threadIdx.x * 2;
        if (i0 >= n) continue;
        unsigned c[2];
        if (i0 + 2 <= n) {
            const unsigned v = *reinterpret_cast<const unsigned *>(code + i0);
            c[0] = v & 0xffffu;
            c[1] = v >> 16;
        } else {
            c[0] = code[i0];
            c[1] = DT_C_MINUS1;
        }
        double id[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            id[k] = -1.0;
            if (c[k] == DT_C_MINUS1) continue;
            const int i = (int)(i0 + k);
            const int t = i / DT_T - (int)((c[k] & DT_C_PREV) != 0);
            const unsigned v = c[k] & DT_C_INDEX;
            const long long wf = (long long)blockIdx.x * 1024 + h * 512 + (threadIdx.x & ~63) * 2;     // the wave's 128 consecutive positions
            const int b = db_bucket_wave(boff, nb, (int)wf, (int)min(wf + 127, (long long)n - 1), i);
            const unsigned rb = runbase[b];
            if (c[k] & DT_C_EXTRA) id[k] = (double)((long long)(runbase[b + 1] - rb) - 1 + (long long)(preE[t] + v - extbase[b]));
            else id[k] = (double)(preR[t] + v - rb);
        }
        if (i0 + 2 <= n && al) *reinterpret_cast<double2 *>(lab + i0) = make_double2(id[0], id[1]);
        else {
            lab[i0] = id[0];
            if (i0 + 1 < n) lab[i0 + 1] = id[1];
        }
    }
}

// one bucket: no scan launch — every workgroup sums the counts it needs itself: whole groups of DT_GRP tiles from the group sums
// the tile kernel accumulated, single tiles inside its own group.  `grp_next` is the group array of the NEXT call (the two
// alternate): workgroup 0 clears it.
#ifndef DT_FTPB
#define DT_FTPB 4                            // tiles per workgroup of dbt_finish1: one prefix prologue serves them all
#endif
__global__ __launch_bounds__(256) void dbt_finish1(const unsigned short *__restrict__ code, double *__restrict__ lab, int n,
                                                   const int *__restrict__ xlab, const unsigned *__restrict__ aggR,
                                                   const unsigned *__restrict__ aggE, int nt, const unsigned *__restrict__ grp,
                                                   unsigned *__restrict__ grp_next, int ng_clear, long long *__restrict__ last_id,
                                                   long long id_base, int use_id_base,
                                                   unsigned *__restrict__ flags, volatile unsigned *host, unsigned seq) {
    __shared__ unsigned red[4][6];
    __shared__ unsigned pR[DT_FTPB + 1], pE[DT_FTPB + 1];      // [j]: runs / extra sub-runs in front of tile (tile0 - 1 + j)
    __shared__ unsigned ownR[DT_FTPB], ownE[DT_FTPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x * DT_FTPB;         // the first of this workgroup's tiles
    dt_signal_host(flags, host, seq);
    const int ng = (nt + DT_GRP - 1) / DT_GRP;
    if (blockIdx.x == 0)
        for (int i = tid; i < ng_clear; i += 256) {
            grp_next[i] = 0;
            grp_next[DT_GRPMAX + i] = 0;
        }
    const long long i0 = (long long)tile * DT_T;
    const long long i1 = i0 + (long long)DT_FTPB * DT_T < (long long)n ? i0 + (long long)DT_FTPB * DT_T : (long long)n;
    // two pairs of points per thread and trip, pair h of thread t at trip*1024 + h*512 + 2t: a store instruction of a wave writes one
    // contiguous KB of labels (DT_T is even: a pair never lies in two tiles).  The codes of every trip are fetched up front, so their
    // latency rides behind the prefix sums below.
    constexpr int NIT = (DT_FTPB * DT_T + 1023) / 1024;
    unsigned cv[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; it++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long i = i0 + it * 1024 + h * 512 + tid * 2;
            cv[it][h] = 0xffffffffu;
            if (i + 2 <= i1) cv[it][h] = *reinterpret_cast<const unsigned *>(code + i);
            else if (i < i1) cv[it][h] = (unsigned)code[i] | 0xffff0000u;
        }
    // A: tiles < tile-1, B: tile-1, T: all
    const int gA = (tile - 1) / DT_GRP;            // group of tile-1 (tile 0: no tile before it)
    unsigned rA = 0, rB = 0, rT = 0, eA = 0, eB = 0, eT = 0;
    for (int g = tid; g < ng; g += 256) {
        const unsigned r = grp[g], e = grp[DT_GRPMAX + g];
        rT += r;
        eT += e;
        if (tile > 0 && g < gA) { rA += r; eA += e; }
    }
    if (tile > 0) {
        const int i = gA * DT_GRP + tid;           // the tiles of group gA in front of tile-1, and tile-1 itself
        if (tid < DT_GRP && i <= tile - 1) {
            const unsigned r = aggR[i], e = aggE[i];
            if (i < tile - 1) { rA += r; eA += e; }
            else { rB = r; eB = e; }
        }
    }
    if (tid < DT_FTPB) {                           // this workgroup's own tiles: fetched beside the sums, not one after the other in the chain below
        const int t = tile + tid;
        ownR[tid] = t < nt ? aggR[t] : 0u;
        ownE[tid] = t < nt ? aggE[t] : 0u;
    }
    unsigned v[6] = {rA, rB, rT, eA, eB, eT};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        for (int d = 32; d > 0; d >>= 1) v[k] += __shfl_xor(v[k], d);
        if (lane == 0) red[wave][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (tid == 0) {                                // exclusive prefixes of tile-1, tile, tile+1, ...: a short serial chain
        unsigned r = v[0], e = v[3];
        pR[0] = r;
        pE[0] = e;
        r += v[1];
        e += v[4];
        for (int j = 1; j <= DT_FTPB; j++) {
            pR[j] = r;
            pE[j] = e;
            r += ownR[j - 1];
            e += ownE[j - 1];
        }
    }
    __syncthreads();
    const long long R1 = use_id_base ? id_base : (long long)v[2] - 1;     // ids of the extra sub-runs continue from here (:112-122)
    if (blockIdx.x == 0 && tid == 0 && last_id) last_id[0] = R1 + (long long)v[5];
    const bool al = (((size_t)lab) & 15) == 0;
#pragma unroll
    for (int it = 0; it < NIT; it++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long i = i0 + it * 1024 + h * 512 + tid * 2;
            if (i >= i1) continue;
            const int tl = (int)((i - i0) / DT_T);
            const unsigned c[2] = {cv[it][h] & 0xffffu, cv[it][h] >> 16};
            double id[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                id[k] = -1.0;
                if (c[k] == DT_C_MINUS1) continue;
                const int j = tl + 1 - (int)((c[k] & DT_C_PREV) != 0);     // the owner tile's prefix (the point's own tile, or the one before it)
                const unsigned v = c[k] & DT_C_INDEX;
                id[k] = (c[k] & DT_C_LITERAL) ? (double)xlab[i + k] : (c[k] & DT_C_EXTRA) ? (double)(R1 + (long long)(pE[j] + v)) : (double)(pR[j] + v);
            }
            if (i + 2 <= n && al) *reinterpret_cast<double2 *>(lab + i) = make_double2(id[0], id[1]);
            else {
                lab[i] = id[0];
                if (i + 1 < n) lab[i + 1] = id[1];
            }
        }
}
