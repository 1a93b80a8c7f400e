// Binned read-depth histogram for gfx950 (MI355X).
//
// Replaces the per-read update_coverage loop of the reference (tiddit_coverage.pyx:48-74 driven by
// __main__.py:229-242 and tiddit_signal.pyx:169-182).  Bit-exactness: every contribution of the
// reference is double(float32(bases)/float32(den)); those quotients are taken from host-built
// IEEE float32 tables and added as exact 2^-S fixed-point int64, so the sum is order independent
// and equals the reference's float64 sum bit for bit.
//
// Kernel shape (HBM-bound streaming, 11 B/read, no MFMA):
//   * one workgroup = 4 waves walks a contiguous chunk of COV_READS_PER_BLOCK coordinate-sorted
//     reads, 8 reads per lane per step (16-byte coalesced loads of start/end/flag, 8 B of mapq);
//   * each lane folds its reads' contributions into two registers (bin K and K+1, K = first bin of
//     its first read); a wavefront segmented reduction over runs of equal K (sorted input => long
//     runs) merges the 64 lanes, so only run-tail lanes touch memory;
//   * those few partial sums go to an LDS-staged per-workgroup window of int64 bins (ds_add_u64),
//     anything outside the window (unsorted input, very long reads) straight to HBM atomics;
//   * the window is spilled once with coalesced 64-bit global atomics, zero bins skipped.
// A final elementwise pass turns int64 accumulators into the float64 bins.
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_coverage;
const char *const tdt_variant_coverage = ""
#ifdef COV_EXP_LOADONLY
    " COV_EXP_LOADONLY"
#endif
#ifdef COV_EXP_NOATOMIC
    " COV_EXP_NOATOMIC"
#endif
#ifdef COV_EXP_NOSCAN
    " COV_EXP_NOSCAN"
#endif
#ifdef COV_M1X_NOK
    " COV_M1X_NOK"
#endif
#ifdef COV_M1X_NOLAST
    " COV_M1X_NOLAST"
#endif
#ifdef COV_M1X_NOSPILL
    " COV_M1X_NOSPILL"
#endif
#ifdef COV_M1X_SPILLKIND
    " COV_M1X_SPILLKIND"
#endif
#ifdef COV_M1_SCANK
    " COV_M1_SCANK"
#endif
#ifdef COV_M1_SKIPZERO
    " COV_M1_SKIPZERO"
#endif
#ifdef COV_M1_SPILL_R4
    " COV_M1_SPILL_R4"
#endif
#ifdef COV_MIN_WAVES
    " COV_MIN_WAVES"
#endif
#ifdef COV_MIN_WAVES1
    " COV_MIN_WAVES1"
#endif
#ifdef COV_MIN_WAVES1_4
    " COV_MIN_WAVES1_4"
#endif
#ifdef COV_MIN_WAVES4
    " COV_MIN_WAVES4"
#endif
#ifdef COV_PF2
    " COV_PF2"
#endif
#ifdef COV_RPL
    " COV_RPL"
#endif
#ifdef COV_RPL1
    " COV_RPL1"
#endif
#ifdef COV_SPARES1
    " COV_SPARES1"
#endif
#ifdef COV_WIN
    " COV_WIN"
#endif
#ifdef COV_WIN1
    " COV_WIN1"
#endif
    ;

#include "tdt_common.h"

#include <algorithm>
#include <thread>

#define COV_THREADS 256
#ifndef COV_RPL
#define COV_RPL 8                                  // reads per lane per step (4 or 8)
#endif
#ifndef COV_RPL1
#define COV_RPL1 4                                 // the same in the small-bin flavour (cov_accumulate MODE 1)
#endif
#define COV_READS_PER_BLOCK 16384                  // consecutive reads per workgroup
#ifndef COV_WIN
#define COV_WIN 2048                               // LDS window, int64 bins (16 KiB)
#endif
#define COV_LUT_LDS_MAX 1024                       // LUT entries kept in LDS (bin_size < 1024)
#define COV_PUSH_CHUNK (4u << 20)                  // reads per staged host chunk
#define COV_STATUS_BYTES (128 + COV_KEPT_SLOTS * 128)
#define COV_KEPT_SLOTS 512                         // kept-read counters, one 128-byte line each

// one contig's slice of a launch
struct CovItem {
    const int32_t *start;
    const int32_t *end;
    const uint8_t *mapq;
    const uint16_t *flag;
    const unsigned long long *packed;    // packed 8-byte records instead of the four arrays (`end` then only serves escapes; may be null)
    unsigned long long n;
    unsigned long long *acc;             // this contig's accumulators
    const unsigned long long *lut_end;   // [bin_size+1] fixed-point float32(b)/float32(end_bin_size)
    int nbins;
    int aligned;                         // all four arrays vector-load aligned
    unsigned first_block;                // first workgroup of this item inside a multi-contig launch
    unsigned binned;                     // `packed` holds cov_bin_record records (start and end then serve the literal path)
};

struct CovParams {
    CovItem it;                          // the item of a single-contig launch
    const CovItem *items;                // multi-contig launch: n_items descriptors in device memory
    int n_items;
    int bin_size;
    unsigned magic;           // floor(x / bin_size) = mulhi(x, magic) >> shift for 0 <= x < 2^31
    int shift;                // -1: bin_size == 1
    int min_q;
    int margin;               // bins: re-base the LDS window when a tile's last read starts this close to its end
    unsigned xmax, m15;       // MODE 1: floor(x / bin_size) == (x * m15) >> k15 for 0 <= x < xmax (<= 2^15), 24-bit multiply
    int k15;
    const unsigned long long *lut_main;  // [bin_size+1] fixed-point float32(b)/float32(bin_size)
    unsigned long long one;              // 1.0 in fixed point
    int *status;                         // [0] |= 1 on range error
    unsigned long long *kept;
};

__device__ __forceinline__ int cov_div(int x, unsigned magic, int shift) {
    return shift < 0 ? x : (int)(__umulhi((unsigned)x, magic) >> shift);
}

// ---- wavefront primitives (DPP: pure VALU cross-lane moves, no LDS crossbar) -------------------
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {  // lanes without a source read 0
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}

// One step of an in-place inclusive wave scan on three 64-bit values at once:
//   x += dpp(x)   as   v_add_co_u32_dpp lo / v_addc_co_u32_dpp hi   (2 instructions per value).
// bound_ctrl:0 makes lanes without a source add 0; rows disabled by row_mask keep their value
// because the add is in place.  The three values are interleaved so that every DPP read of a VGPR
// is >= 4 instructions behind the VALU write of that VGPR (the DPP read-after-write hazard needs 2
// wait states, which inline asm does not get from the compiler).
// The whole scan is ONE asm statement that opens with the wait states: the compiler can then place nothing (a v_mov that materialises an
// operand, a v_cndmask of the caller) between the s_nop and the first DPP read, nor between two steps.
#define COV_DPP_SHR1 "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define COV_DPP_SHR2 "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define COV_DPP_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define COV_DPP_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define COV_DPP_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define COV_DPP_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
#define COV_SCAN_ALL(STEP) \
    "s_nop 1\n\t" STEP(COV_DPP_SHR1) STEP(COV_DPP_SHR2) STEP(COV_DPP_SHR4) STEP(COV_DPP_SHR8) STEP(COV_DPP_BC15) STEP(COV_DPP_BC31) "s_nop 1"
#define COV_SCAN_STEP(CTRL)                                                                     \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\t"                                             \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL "\n\t"                                       \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " CTRL "\n\t"                                             \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " CTRL "\n\t"                                       \
    "v_add_co_u32_dpp %4, vcc, %4, %4 " CTRL "\n\t"                                             \
    "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " CTRL "\n\t"

__device__ __forceinline__ void wave_scan3_u64(unsigned long long &a0, unsigned long long &a1, unsigned long long &a2) {
    unsigned l0 = (unsigned)a0, h0 = (unsigned)(a0 >> 32), l1 = (unsigned)a1, h1 = (unsigned)(a1 >> 32),
             l2 = (unsigned)a2, h2 = (unsigned)(a2 >> 32);
    asm volatile(COV_SCAN_ALL(COV_SCAN_STEP) : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2)::"vcc", "memory");   // (the values were just written by plain VALU)
    a0 = ((unsigned long long)h0 << 32) | l0;
    a1 = ((unsigned long long)h1 << 32) | l1;
    a2 = ((unsigned long long)h2 << 32) | l2;
}

// Lane predicates as 64-bit masks in SGPRs (one bit per lane), so that their algebra, the kept-read count (s_bcnt1) and the "any lane
// needs the literal path" flag run on the scalar unit.  (The compiler's own ballot of a combined predicate materialises it in a VGPR
// and compares again: two vector instructions per use.)
__device__ __forceinline__ unsigned long long cov_mask_lt(unsigned a, unsigned lim_uniform) {   // a < lim
    unsigned long long m;
    asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "s"(lim_uniform), "v"(a));
    return m;
}
__device__ __forceinline__ unsigned long long cov_mask_eq0(unsigned a) {
    unsigned long long m;
    asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
    return m;
}
__device__ __forceinline__ unsigned long long cov_mask_ne0(unsigned a) {
    unsigned long long m;
    asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
    return m;
}
__device__ __forceinline__ unsigned cov_select(unsigned long long m, unsigned if_set, unsigned if_clear) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}

// the same for two values (each DPP read is still three instructions behind the write of its register)
#define COV_SCAN2_STEP(CTRL)                                                                    \
    "v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\t"                                             \
    "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL "\n\t"                                       \
    "v_add_co_u32_dpp %2, vcc, %2, %2 " CTRL "\n\t"                                             \
    "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " CTRL "\n\t"

__device__ __forceinline__ void wave_scan2_u64(unsigned long long &a0, unsigned long long &a1) {
    unsigned l0 = (unsigned)a0, h0 = (unsigned)(a0 >> 32), l1 = (unsigned)a1, h1 = (unsigned)(a1 >> 32);
    asm volatile(COV_SCAN_ALL(COV_SCAN2_STEP) : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1)::"vcc", "memory");   // (the values were just written by plain VALU)
    a0 = ((unsigned long long)h0 << 32) | l0;
    a1 = ((unsigned long long)h1 << 32) | l1;
}

// four 32-bit values at once (each DPP read is three instructions behind the write of its register): the window resolve of MODE 1
#define COV_SCAN4I_STEP(CTRL)                                                                   \
    "v_add_u32_dpp %0, %0, %0 " CTRL "\n\t"                                                     \
    "v_add_u32_dpp %1, %1, %1 " CTRL "\n\t"                                                     \
    "v_add_u32_dpp %2, %2, %2 " CTRL "\n\t"                                                     \
    "v_add_u32_dpp %3, %3, %3 " CTRL "\n\t"

__device__ __forceinline__ void wave_scan4_i32(int &p0, int &p1, int &p2, int &p3) {
    asm volatile(COV_SCAN_ALL(COV_SCAN4I_STEP) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)::"memory");
}

// 16 bytes at an LDS byte address held in a register (ds_read_b128, no generic-pointer arithmetic)
__device__ __forceinline__ ulonglong2 cov_lds_read128(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned long long cov_v2u64 __attribute__((ext_vector_type(2)));
    const cov_v2u64 v = *reinterpret_cast<const cov_v2u64 __attribute__((address_space(3))) *>(addr);
    return make_ulonglong2(v.x, v.y);
#else
    (void)addr;
    return make_ulonglong2(0, 0);
#endif
}

__device__ __forceinline__ unsigned long long cov_lds_read64(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const unsigned long long __attribute__((address_space(3))) *>(addr);
#else
    return addr;
#endif
}
__device__ __forceinline__ void cov_lds_add64(unsigned addr, unsigned long long v) {           // ds_add_u64
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long __attribute__((address_space(3))) *>(addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    (void)addr; (void)v;
#endif
}
// r + (the lane's bit of mask): one v_addc
__device__ __forceinline__ unsigned cov_add_bit(unsigned r, unsigned long long mask) {
    unsigned out;
    unsigned long long cout;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(out), "=s"(cout) : "v"(r), "s"(mask));
    return out;
}

// out-of-window contributions (unsorted input, reads longer than the window): plain HBM atomics.
// noinline keeps the LDS path a real ds_add_u64 instead of a flat atomic on a selected pointer.
__device__ __noinline__ void cov_global_add(unsigned long long *acc, int bin, unsigned long long v) {
    atomicAdd(&acc[bin], v);
}

template <int RPL>
struct CovTile {
    int s[RPL], e[RPL];
    unsigned mq[RPL / 4];   // 4 mapq bytes per word
    unsigned fl[RPL / 2];   // 2 flags per word
};

template <int RPL>
__device__ __forceinline__ CovTile<RPL> cov_load(const CovItem &P, unsigned long long idx, unsigned long long r1) {
    CovTile<RPL> t;
    if (P.aligned && idx + RPL <= r1) {
#pragma unroll
        for (int k = 0; k < RPL / 4; k++) {
            const int4 s4 = *reinterpret_cast<const int4 *>(P.start + idx + 4 * k);
            const int4 e4 = *reinterpret_cast<const int4 *>(P.end + idx + 4 * k);
            t.s[4 * k] = s4.x; t.s[4 * k + 1] = s4.y; t.s[4 * k + 2] = s4.z; t.s[4 * k + 3] = s4.w;
            t.e[4 * k] = e4.x; t.e[4 * k + 1] = e4.y; t.e[4 * k + 2] = e4.z; t.e[4 * k + 3] = e4.w;
        }
        if (RPL == 8) {
            const uint2 m2 = *reinterpret_cast<const uint2 *>(P.mapq + idx);
            const uint4 f4 = *reinterpret_cast<const uint4 *>(P.flag + idx);
            t.mq[0] = m2.x; t.mq[RPL / 4 - 1] = m2.y;
            t.fl[0] = f4.x; t.fl[1] = f4.y; t.fl[RPL / 2 - 2] = f4.z; t.fl[RPL / 2 - 1] = f4.w;
        } else {
            t.mq[0] = *reinterpret_cast<const unsigned *>(P.mapq + idx);
            const uint2 f2 = *reinterpret_cast<const uint2 *>(P.flag + idx);
            t.fl[0] = f2.x; t.fl[1] = f2.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < RPL / 4; k++) t.mq[k] = 0;
#pragma unroll
        for (int k = 0; k < RPL / 2; k++) t.fl[k] = 0;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const bool ok = idx + j < r1;
            t.s[j] = ok ? P.start[idx + j] : 0;
            t.e[j] = ok ? P.end[idx + j] : 1;
            t.mq[j / 4] |= (ok ? (unsigned)P.mapq[idx + j] : 0u) << (8 * (j & 3));
            t.fl[j / 2] |= (ok ? (unsigned)P.flag[idx + j] : 0x4u) << (16 * (j & 1));  // padding lanes look unmapped
        }
    }
    return t;
}

// Packed alignment record (8 B instead of 11): what the ingest kernel can write next to the field arrays.
//   low word  = reference_start (int32)
//   high word = span:24 | min(mapq,63):6 | unmapped(0x4):1 | duplicate(0x400):1     span = reference_end - reference_start;
//               span 0xffffff = escape: the true end is read from the `end` array (reads spanning >= 16 Mb)
template <int RPL>
struct CovTileP {
    unsigned long long w[RPL];
};

template <int RPL>
__device__ __forceinline__ CovTileP<RPL> cov_load_packed(const CovItem &P, unsigned long long idx, unsigned long long r1) {
    CovTileP<RPL> t;
    if (P.aligned && idx + RPL <= r1) {
#pragma unroll
        for (int k = 0; k < RPL / 2; k++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(P.packed + idx + 2 * k);
            t.w[2 * k] = v.x;
            t.w[2 * k + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RPL; j++) t.w[j] = idx + j < r1 ? P.packed[idx + j] : (1ull << 62);   // padding lanes look unmapped
    }
    return t;
}

__global__ void cov_pack(const int32_t *__restrict__ start, const int32_t *__restrict__ end, const uint8_t *__restrict__ mapq,
                         const uint16_t *__restrict__ flag, unsigned long long n, unsigned long long *__restrict__ out) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = cov_pack_record(start[i], end[i], mapq[i], flag[i]);
}

__global__ void cov_pack_binned(const int32_t *__restrict__ start, const int32_t *__restrict__ end, const uint8_t *__restrict__ mapq,
                                const uint16_t *__restrict__ flag, unsigned long long n, int nbins, unsigned z, unsigned magic, int shift, int mode1,
                                unsigned long long *__restrict__ out) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = cov_bin_record(start[i], end[i], mapq[i], flag[i], nbins, z, magic, shift, mode1 != 0);
}

// MODE 0 (many reads per bin, e.g. --cov at 500 bp): contributions to bins K, K+1, K+2 are folded into three
//   registers per lane and merged across the wave by a prefix scan over runs of equal K.
// MODE 1 (few reads per bin, e.g. --sv at 50 bp, where a 150-bp read covers 3-5 bins and neighbouring lanes hardly
//   ever share K): the "+1.0 for every bin strictly inside the read" (tiddit_coverage.pyx:71-72) becomes a
//   difference pair — +1 at first_bin+1, -1 at last_bin — kept in the top 16 bits of the same 64-bit LDS window word
//   that holds the bin's partial sums (low 48 bits), so a read of ANY length costs one ds_add_u64 for its last bin
//   (partial - 2^48, read ready-made from an LDS table) plus register adds for its first bin (partial) and
//   first_bin+1 (+2^48); the window is resolved by one prefix sum over the difference counts when it is spilled.
//   Everything a read contributes is self-masked through its table index (entry 0 of every table is 0), so no
//   predicate outlives the read that produced it.  Only reads whose bins all lie inside the window and before the
//   contig's last bin take this path (the differences must cancel inside the window); the rest are replayed literally.
#define COV_DBIT 48
#define COV_LOWMASK ((1ull << COV_DBIT) - 1ull)
#ifndef COV_WIN1
#define COV_WIN1 2048                                 // MODE 1 window (16 KiB: eight workgroups per CU)
#endif
#define COV_DQMAX 256                               // MODE 1: a register-path read ends at most this many bins after K
#ifndef COV_SPARES1
#define COV_SPARES1 64                              // MODE 1: spare words behind the window, one per lane (see the binned loop)
#endif

#ifndef COV_MIN_WAVES
#define COV_MIN_WAVES 4                            // waves per SIMD the register allocation must allow (MODE 0)
#endif
#ifndef COV_MIN_WAVES4
#define COV_MIN_WAVES4 3                           // ... MODE 0 fed from four arrays: at 4 it spills 14 registers to scratch
#endif
#ifndef COV_MIN_WAVES1
#define COV_MIN_WAVES1 6                           // ... MODE 1 on 8-byte records (80 registers, nothing spilled; the one-pass window resolve of round 5 would take 81)
#endif
#ifndef COV_MIN_WAVES1_4
#define COV_MIN_WAVES1_4 4                         // ... MODE 1 fed from four arrays (at 6 it spills 5 registers)
#endif
// REC: the record layout of the stream — 0: four arrays (start, end, mapq, flag), 1: 8-byte packed records (cov_pack_record),
//      2: 8-byte BINNED records (cov_bin_record: first bin, table indices and shape precomputed for this histogram's bin size)
template <bool LDS_LUT, int MODE, bool Z1, int RPL, int REC>
__global__ __launch_bounds__(COV_THREADS, MODE == 1 ? (REC ? COV_MIN_WAVES1 : COV_MIN_WAVES1_4) : (REC ? COV_MIN_WAVES : COV_MIN_WAVES4)) void cov_accumulate(CovParams P) {
    constexpr bool PACKED = REC != 0;                  // 8-byte records in I.packed
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];
    // everything lives in the dynamic region (a static __shared__ in front of it would shift its
    // base off 16-byte alignment): [0..11] scratch (4 scan words, then the bin of every tile's last read), [12..) window
    // (+2 spare words), then the LUT pair
    // (main table, then the contig's end-bin table, each bin_size+1 entries); MODE 1 adds three tables behind them
    constexpr int WIN = MODE == 1 ? COV_WIN1 : COV_WIN;
    constexpr int TILE = COV_THREADS * RPL;            // reads per workgroup step
    int *s_wsum = reinterpret_cast<int *>(smem);
    int *s_tbin = s_wsum + 4;                          // [COV_READS_PER_BLOCK / TILE] <= 16 entries
    // (MODE 1: 32 more ints in front of the window — the per-(row, wave) difference totals of the window resolve)
    int *s_rtot = reinterpret_cast<int *>(smem + 12);
    unsigned long long *win = smem + 12 + (MODE == 1 ? 16 : 0);
    unsigned long long *lutS = win + WIN + (MODE == 1 ? COV_SPARES1 : 2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // which contig does this workgroup belong to (wave-uniform: scalar loads + scalar binary search)
    CovItem I = P.it;
    unsigned blk = blockIdx.x;
    if (P.n_items > 0) {
        int lo = 0, hi = P.n_items;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (P.items[mid].first_block <= blk) lo = mid;
            else hi = mid;
        }
        I = P.items[lo];
        blk -= I.first_block;
    }
    const unsigned long long r0 = (unsigned long long)blk * COV_READS_PER_BLOCK;
    const unsigned long long r1 = min(I.n, r0 + COV_READS_PER_BLOCK);
    const unsigned z = (unsigned)P.bin_size;
    const int last_bin = I.nbins - 1;
    auto div = [&](int x) { return Z1 ? x : (int)(__umulhi((unsigned)x, P.magic) >> P.shift); };

    // first tile's loads go out before the LDS set-up
    auto load_tile = [&](unsigned long long idx) {
        if constexpr (PACKED) return cov_load_packed<RPL>(I, idx, r1);
        else return cov_load<RPL>(I, idx, r1);
    };
    auto cur = load_tile(r0 + (unsigned long long)tid * RPL);
#ifdef COV_PF2           // measurement variant: two tiles in flight per wave instead of one (MODE 1)
    constexpr bool PF2 = MODE == 1;
    auto ahead = cur;
    if (PF2 && r0 + TILE < r1) ahead = load_tile(r0 + TILE + (unsigned long long)tid * RPL);
#endif

    for (int i = tid; i < WIN + 2; i += COV_THREADS) win[i] = 0;
    if (LDS_LUT) {
        for (unsigned i = tid; i <= z; i += COV_THREADS) {
            const unsigned long long v = P.lut_main[i];
            lutS[i] = v;
            lutS[i + z + 1] = I.lut_end[i];
            if (MODE == 0 && !Z1) {
                // pairA[i] = (v, 0), pairB[i] = (0, v) as 16-byte entries (see the tabled loop below)
                unsigned long long *pr = lutS + 2 * (z + 1);
                pr[2 * i] = v;
                pr[2 * i + 1] = 0;
                pr[2 * (z + 1) + 2 * i] = 0;
                pr[2 * (z + 1) + 2 * i + 1] = v;
            }
            if (MODE == 1) {
                // tabA[i] = (v, 0), tabB[i] = (0, v): one 8-byte read yields the first-bin partial already routed to
                // bin K (first word) or K+1 (second word); tabL[i] = v - 2^48: the last bin's partial with its -1;
                // tabL[z+1] = 0 is where masked reads point
                unsigned long long *tabA = lutS + 2 * (z + 1), *tabB = tabA + (z + 1), *tabL = tabB + (z + 1);
                tabA[i] = v & 0xffffffffull;
                tabB[i] = v << 32;
                tabL[i] = v - (1ull << COV_DBIT);
                if (i == 0) tabL[z + 1] = 0;
            }
        }
    }
    // window base: the bin of the chunk's first read; every thread derives it from the same (scalar) load
    auto bin_of_read = [&](unsigned long long idx) {
        if constexpr (REC == 2) return (int)((unsigned)I.packed[idx] >> 2);     // (clamped to the contig's bins when the record was made)
        int s = PACKED ? (int)(unsigned)I.packed[idx] : I.start[idx];
        s = s < 0 ? 0 : s;
        const int b = div(s);
        return b < I.nbins ? b : I.nbins - 1;
    };
    // bins of the tiles' last reads (for the window re-base below): one parallel round of loads instead of a
    // dependent scalar load in every step
    if (tid < COV_READS_PER_BLOCK / TILE) {
        const unsigned long long tl = min(r0 + (unsigned long long)(tid + 1) * TILE, r1) - 1;
        s_tbin[tid] = bin_of_read(tl);
    }
    int base = bin_of_read(r0);
    __syncthreads();
    // LUT[i] for i <= z: float32(i)/float32(z); LUT[z+1+i]: float32(i)/float32(end_bin_size).  LUT[0] == 0.
    auto lut = [&](unsigned i) -> unsigned long long {
        if (LDS_LUT) return lutS[i];
        return i <= z ? P.lut_main[i] : I.lut_end[i - z - 1];
    };

    // plain (difference-free) contribution, any bin
    auto contribute = [&](int bin, unsigned long long v) {
        const unsigned off = (unsigned)(bin - base);
#ifdef COV_EXP_NOATOMIC  // measurement variant (tools/build_variant.sh): everything but the LDS atomics — DESIGN.md §3.1 ceilings
        if (v == 0x1234567ull) win[0] = v;
#else
        if (off < (unsigned)WIN) atomicAdd(&win[off], v);  // ds_add_u64 (bins past the contig end only ever see +x and -x)
        else if ((unsigned)bin <= (unsigned)last_bin) cov_global_add(I.acc, bin, v);
#endif
    };

    // the literal per-read update (tiddit_coverage.pyx:50-72) for reads the register path cannot hold
    auto slow_read = [&](int s, int e) {
        const int fb = div(s);
        const int eb = div(e - 1);
        if (fb == eb) {
            contribute(fb, lut((unsigned)(e - s)));
        } else {
            contribute(fb, lut((unsigned)(fb + 1) * z - (unsigned)s));
            const unsigned bl = (unsigned)(e - 1) - (unsigned)eb * z;
            contribute(eb, lut(eb < last_bin ? bl : bl + z + 1));
            for (int b = fb + 1; b < eb; b++) contribute(b, P.one);
        }
    };

    // spill the LDS window to the accumulators (coalesced 64-bit global atomics, zero bins skipped) and clear it
    auto spill = [&]() {
        __syncthreads();
#if !defined(COV_M1_SPILL_R4)
        if constexpr (MODE == 1) {
            // ONE pass resolves the difference counts and spills (round 5; COV_M1_SPILL_R4 keeps round 4's two passes for A/B runs).
            // Thread t owns the words t, t + 256, ... (row k = words [256 k, 256 k + 256)): every LDS access of the pass is a wave's
            // contiguous 512 bytes (round 4 gave a thread eight CONSECUTIVE words — a 64-byte lane stride, four lanes per bank — read them
            // twice and wrote them twice).  The prefix over the window = per row a wave scan (the eight rows' 16-bit difference counts
            // packed two to a register: a workgroup holds at most 2^14 reads, so no partial sum leaves [-2^14, 2^14] and the fields never
            // borrow beyond what the extraction undoes) + the 32 (row, wave) totals, scanned by every wave for itself.
            constexpr int PER = WIN / COV_THREADS;
            static_assert(PER == 8 && COV_THREADS == 256, "the packed resolve is written for eight rows of 256 words");
            unsigned long long w[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) w[k] = win[k * COV_THREADS + tid];
            int pk[PER / 2];
#pragma unroll
            for (int j = 0; j < PER / 2; j++)
                pk[j] = (int)((long long)w[2 * j] >> COV_DBIT) + (int)((unsigned)((long long)w[2 * j + 1] >> COV_DBIT) << 16);
            wave_scan4_i32(pk[0], pk[1], pk[2], pk[3]);
            int incl[PER];
#pragma unroll
            for (int j = 0; j < PER / 2; j++) {
                incl[2 * j] = (int)(short)(pk[j] & 0xffff);
                incl[2 * j + 1] = (pk[j] - incl[2 * j]) >> 16;
            }
            const int wv = tid >> 6;
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < PER; k++) s_rtot[k * 4 + wv] = incl[k];
            }
            __syncthreads();
            int tot = lane < 4 * PER ? s_rtot[lane] : 0, z1 = 0, z2 = 0, z3 = 0;
            const int own = tot;
            wave_scan4_i32(tot, z1, z2, z3);
            const int excl = tot - own;                       // lane 4 k + v: the differences in front of row k's share of wave v
            const unsigned sh = (unsigned)__builtin_ctzll(P.one);
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int run = incl[k] + __builtin_amdgcn_readlane(excl, k * 4 + wv);
                const unsigned long long v = (w[k] & COV_LOWMASK) + ((unsigned long long)(long long)run << sh);
                const int i = k * COV_THREADS + tid;
                if (w[k]) win[i] = 0;
                if (v) {
#ifdef COV_M1X_NOSPILL   // measurement variant: what the 60 M global atomics of the 50-bp flavour cost
                    if (v == 0x1234567ull)
#endif
#if defined(COV_M1X_SPILLKIND) && COV_M1X_SPILLKIND == 1      // MEASUREMENT ONLY (races at the borders): plain load / add / store
                    if (base + i <= last_bin) I.acc[base + i] += v;
#elif defined(COV_M1X_SPILLKIND) && COV_M1X_SPILLKIND == 2    // MEASUREMENT ONLY: atomics executed in the XCD's own L2 (workgroup scope)
                    if (base + i <= last_bin) __hip_atomic_fetch_add(&I.acc[base + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif defined(COV_M1X_SPILLKIND) && COV_M1X_SPILLKIND == 3    // MEASUREMENT ONLY: stores
                    if (base + i <= last_bin) I.acc[base + i] = v;
#else
                    if (base + i <= last_bin) atomicAdd(&I.acc[base + i], v);
#endif
                }
            }
            __syncthreads();                                  // (s_rtot is rewritten by the next spill, the window by the next tile)
            return;
        }
#endif
        if (MODE == 1) {
            // resolve the difference counts: thread t owns window entries [PER*t, PER*t + PER)
            constexpr int PER = WIN / COV_THREADS;
            int d = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) d += (int)((long long)win[tid * PER + k] >> COV_DBIT);
            int incl = d;
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (lane == 63) s_wsum[tid >> 6] = incl;
            __syncthreads();
            int run = incl - d;
            for (int wv = 0; wv < (tid >> 6); wv++) run += s_wsum[wv];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const unsigned long long w = win[tid * PER + k];
                run += (int)((long long)w >> COV_DBIT);
                win[tid * PER + k] = (w & COV_LOWMASK) + (unsigned long long)(long long)run * P.one;
            }
            __syncthreads();
        }
        for (int i = tid; i < WIN; i += COV_THREADS) {
            const unsigned long long v = win[i];
            if (v) {
#ifdef COV_M1X_NOSPILL
                if (MODE != 1 || v == 0x1234567ull)
#endif
                if (base + i <= last_bin) atomicAdd(&I.acc[base + i], v);
                win[i] = 0;
            }
        }
    };

    unsigned nkept = 0;
    unsigned nkept_s = 0;                              // kept reads counted by wave ballots (binned records): the same value in every lane
    bool bad = false;

    for (unsigned long long t0 = r0; t0 < r1; t0 += TILE) {
        // software prefetch: the next tile's loads are in flight while this one is reduced
#ifdef COV_PF2
        auto nxt = cur;
        if constexpr (PF2) {
            nxt = ahead;
            if (t0 + 2 * TILE < r1) ahead = load_tile(t0 + 2 * TILE + (unsigned long long)tid * RPL);
        } else if (t0 + TILE < r1) nxt = load_tile(t0 + TILE + (unsigned long long)tid * RPL);
#else
        auto nxt = cur;
        if (t0 + TILE < r1) nxt = load_tile(t0 + TILE + (unsigned long long)tid * RPL);
#endif

        // window re-base (block-uniform): when this tile's last read starts near the window's end the window is spilled
        // and moved to the previous tile's last read, so sparse streams / small bins stay on the LDS path
        if (t0 != r0) {
            const int step = (int)((t0 - r0) / TILE);
            if (s_tbin[step] + P.margin >= base + WIN) {
                const int nb = s_tbin[step - 1];
                if (nb != base) {
                    spill();
                    __syncthreads();
                    base = nb;
                }
            }
        }

        if constexpr (REC == 2) {
            // ---- binned records (cov_bin_record, tdt_common.h): low word = first_bin << 2 | shape, high word = filter byte | table indices.
            // Everything that depends only on the read and the bin size — the division, the single-/two-bin split, the bases in the
            // first and last bin with the "one short" quirk (tiddit_coverage.pyx:50-63), the validity checks — was done once when the
            // record was written (ingest / tdt_cov_pack_binned_device).  What is left per read: its bin relative to the lane key, the
            // read filter (one range test), two table reads and the three 64-bit adds.  Kept-read counts and the "some read needs the
            // literal path" flag are wave ballots in SGPRs (scalar unit), not per-lane VALU work.
            static_assert(LDS_LUT && !Z1, "binned records exist for 2 <= bin_size < 1024");
            const unsigned K4 = (unsigned)cur.w[0] & ~3u;        // lane key: the first bin of the lane's first read (any K is correct)
            const int K = (int)(K4 >> 2);
            const unsigned ko = (unsigned)(K - base);
            const unsigned minq24 = (unsigned)P.min_q << 24, qlim = (64u - (unsigned)P.min_q) << 24;
            unsigned long long slow_any = 0;
            unsigned long long a0 = 0, a1 = 0, a2 = 0;
#ifdef COV_EXP_LOADONLY  // measurement variant: the loads alone (the ceiling of this access pattern)
            for (int j = 0; j < RPL; j++) a0 += cur.w[j];
            if (a0 == 0x1234567ull) contribute(K, a0);
            cur = nxt;
            continue;
#endif
            if constexpr (MODE == 0) {
                const bool safe = ko < (unsigned)(WIN - 3);
                const unsigned kw = safe ? ko : 0u;
                // LDS byte addresses of the (v, 0) table and of the (0, v) table behind it, in registers: a table read is then
                // `field + table` with the masked case `table` itself (entry 0 = (0, 0)) — no address add behind the select
                const unsigned tA = (unsigned)(size_t)(lutS + 2 * (z + 1)), zb4 = (z + 1u) << 2;
                const unsigned notsafe = safe ? 0u : 8u;
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    const unsigned lo = (unsigned)cur.w[j], hi = (unsigned)(cur.w[j] >> 32);
                    const unsigned d = lo - K4;                  // (first bin - K) << 2 | shape: the register path takes 0, 1 (bin K) and 4, 5 (bin K+1)
                    // __main__.py:231-235 / tiddit_signal.pyx:171-181 as one range test of the filter byte
                    const unsigned long long m_cand = cov_mask_lt(hi - minq24, qlim);
                    const unsigned long long m_fast = m_cand & cov_mask_eq0((d & ~5u) | notsafe);
                    nkept_s += (unsigned)__builtin_popcountll(m_fast);
                    slow_any |= m_cand & ~m_fast;
                    const unsigned tab = __umul24(d & 4u, zb4) + tA;                  // first bin K+1: the (0, v) table
                    const unsigned xa = cov_select(m_fast, (hi & 0x3ff0u) + tab, tA);
                    const ulonglong2 X = cov_lds_read128(xa);
                    const unsigned long long m_multi = m_fast & cov_mask_ne0(d & 1u);
                    const unsigned ya = cov_select(m_multi, (((hi >> 14) & 0x3ffu) << 4) + tab, tA);
                    const ulonglong2 Y = cov_lds_read128(ya);
                    a0 += X.x;
                    a1 += X.y + Y.x;
                    a2 += Y.y;
                }
                if (slow_any) {                                  // wave-uniform; rare: reads of three or more bins, the contig's last bin, unsorted input
                    const unsigned long long idx = t0 + (unsigned long long)tid * RPL;
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        unsigned lo = (unsigned)cur.w[j], hi = (unsigned)(cur.w[j] >> 32);
                        asm volatile("" : "+v"(lo), "+v"(hi));       // recomputed here, not kept alive from the loop above (registers)
                        const unsigned d = lo - K4;
                        if ((hi - minq24 < qlim) && !(safe & ((d & ~5u) == 0u))) {
                            if ((lo & 3u) == COV_BN_INVALID || idx + j >= r1) bad = true;
                            else { nkept++; slow_read(I.start[idx + j], I.end[idx + j]); }
                        }
                    }
                }
                // bin K+2 only ever receives the last-bin part of a two-bin read that starts in bin K+1 — reads behind a bin boundary inside
                // the lane's eight, a few lanes per wave and hardly two of them on one bin: those go to the window directly, and only
                // the sums of bins K and K+1 take the wavefront merge (two values instead of three through the scan)
#ifdef COV_EXP_NOATOMIC
                if (a0 + a1 + a2 == 0x1234567ull) win[0] = a0;
                cur = nxt;
                continue;
#endif
#ifdef COV_EXP_NOSCAN
                if (a0 + a1 + a2 == 0x1234567ull) win[0] = a0;
                if (a2) atomicAdd(&win[kw + 2], a2);
                cur = nxt;
                continue;
#endif
                if (a2) atomicAdd(&win[kw + 2], a2);
                wave_scan2_u64(a0, a1);
                const int Kprev = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHR1, 0xf, 0xf, false);
                const int Knext = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHL1, 0xf, 0xf, false);
                const unsigned long long q0 = dpp_u64<DPP_WAVE_SHR1>(a0), q1 = dpp_u64<DPP_WAVE_SHR1>(a1);
                if (safe && Knext != K) {  // run tail (lane 63 always: it reads ~K)
                    if (a0) atomicAdd(&win[kw], a0);
                    if (a1) atomicAdd(&win[kw + 1], a1);
                }
                if (safe && Kprev != K && lane != 0) {  // run head
                    if (q0) atomicAdd(&win[kw], 0ull - q0);
                    if (q1) atomicAdd(&win[kw + 1], 0ull - q1);
                }
            } else {
                // small bins (MODE 1): difference pairs in the window words, see above; fields bf:8 | bl:8 | bins after the first:8.
                // Predicates are SGPR masks; table and window accesses go through LDS byte addresses held in registers, the masked
                // case being the table's zero entry / the window's spare word.
                const bool safe = ko < (unsigned)(WIN - COV_DQMAX - 2);
                const unsigned kw = safe ? ko : 0u;
                const unsigned notsafe = safe ? 0u : 8u;
                const unsigned tA = (unsigned)(size_t)(lutS + 2 * (z + 1)), zb2 = (z + 1u) << 1;            // tabA, then tabB = tabA + (z+1) entries
                const unsigned tL = tA + ((z + 1u) << 4), tL0 = tL + ((z + 1u) << 3);                        // tabL and its zero entry
                // a masked read adds a zero to a spare word behind the window — one word PER LANE: with a single word the masked lanes of
                // an instruction (filtered reads, ~5 %) all hit one address and the ds_add_u64 serialises over them
                const unsigned wK = (unsigned)(size_t)(win + kw), wSpare = (unsigned)(size_t)(win + WIN + (COV_SPARES1 >= 64 ? lane : 0));
                unsigned d12 = 0, d2 = 0;                        // reads that put their +1 on bin K+1 or K+2 / on K+2
                unsigned long long vL[RPL];
                unsigned woff[RPL];
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    const unsigned lo = (unsigned)cur.w[j], hi = (unsigned)(cur.w[j] >> 32);
                    const unsigned d = lo - K4;
                    const unsigned long long m_cand = cov_mask_lt(hi - minq24, qlim);
                    const unsigned long long m_fast = m_cand & cov_mask_eq0((d & ~5u) | notsafe);
                    nkept_s += (unsigned)__builtin_popcountll(m_fast);
                    slow_any |= m_cand & ~m_fast;
                    const unsigned r4 = d & 4u;                  // first bin = K + 1
                    const unsigned tab = __umul24(r4, zb2) + tA;
                    const unsigned long long vv = cov_lds_read64(cov_select(m_fast, ((hi & 0xffu) << 3) + tab, tA));   // (to bin K, to bin K+1)
                    a0 += (unsigned)vv;
                    a1 += (unsigned)(vv >> 32);
                    const unsigned long long m_multi = m_fast & cov_mask_ne0(d & 1u);
                    d12 = cov_add_bit(d12, m_multi);
                    d2 = cov_add_bit(d2, m_multi & cov_mask_ne0(r4));
                    vL[j] = cov_lds_read64(cov_select(m_multi, (((hi >> 8) & 0xffu) << 3) + tL, tL0));     // the last bin's quotient with the -1 of the pair
                    woff[j] = cov_select(m_multi, (((d >> 2) + ((hi >> 16) & 0xffu)) << 3) + wK, wSpare);
                }
#ifndef COV_M1X_NOLAST   // COV_M1X_*: measurement variants (tools/build_variant.sh) — what each group of LDS operations costs
#pragma unroll
                for (int j = 0; j < RPL; j++) cov_lds_add64(woff[j], vL[j]);
#else
                if (vL[0] + vL[1] + vL[2] + vL[3] + woff[0] + woff[1] + woff[2] + woff[3] == 0x1234567ull) win[0] = 1;
#endif
#ifdef COV_M1X_NOK
                if (a0 + a1 + d12 + d2 == 0x1234567ull) win[0] = a0;
#elif defined(COV_M1_SCANK)
                // equal-K neighbours (2-3 lanes share a 50-bp first bin) merged before they reach the window: inclusive wave scan of the
                // two sums, the run's tail adds P[tail], its head takes P[head - 1] back (exact in modular arithmetic, like MODE 0)
                {
                    unsigned long long s0 = a0, s1 = a1 + ((unsigned long long)(d12 - d2) << COV_DBIT);
                    if (d2) atomicAdd(&win[kw + 2], (unsigned long long)d2 << COV_DBIT);
                    wave_scan2_u64(s0, s1);
                    const int Kprev = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHR1, 0xf, 0xf, false);
                    const int Knext = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHL1, 0xf, 0xf, false);
                    const unsigned long long q0 = dpp_u64<DPP_WAVE_SHR1>(s0), q1 = dpp_u64<DPP_WAVE_SHR1>(s1);
                    if (safe && Knext != K) {
                        atomicAdd(&win[kw], s0);
                        atomicAdd(&win[kw + 1], s1);
                    }
                    if (safe && Kprev != K && lane != 0) {
                        atomicAdd(&win[kw], 0ull - q0);
                        atomicAdd(&win[kw + 1], 0ull - q1);
                    }
                }
#else
                atomicAdd(&win[kw], a0);
                atomicAdd(&win[kw + 1], a1 + ((unsigned long long)(d12 - d2) << COV_DBIT));
#ifdef COV_M1_SKIPZERO
                if (d2) atomicAdd(&win[kw + 2], (unsigned long long)d2 << COV_DBIT);
#else
                atomicAdd(&win[kw + 2], (unsigned long long)d2 << COV_DBIT);
#endif
#endif
                if (slow_any) {
                    const unsigned long long idx = t0 + (unsigned long long)tid * RPL;
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const unsigned lo = (unsigned)cur.w[j], hi = (unsigned)(cur.w[j] >> 32);
                        const unsigned d = lo - K4;
                        if ((hi - minq24 < qlim) && !(safe & ((d & ~5u) == 0u))) {
                            if ((lo & 3u) == COV_BN_INVALID || idx + j >= r1) bad = true;
                            else { nkept++; slow_read(I.start[idx + j], I.end[idx + j]); }
                        }
                    }
                }
            }
            cur = nxt;
            continue;
        }
        int sv[RPL], ev[RPL];
        unsigned mq[RPL], fl[RPL];
        constexpr unsigned FMASK = PACKED ? 0x3u : 0x404u;       // unmapped / duplicate bits of fl[]
        // The small-bin flavour never folds a read of the escape span (16 Mb) into registers: it reaches the literal path, which looks its
        // true end up there (ESC_LATE: 2 % of that launch).  The other flavours patch the ends up front — the same change costs the
        // 500-bp flavour, which has no register to spare, 21 spilled registers and a quarter of its speed.
        constexpr bool ESC_LATE = PACKED && MODE == 1 && LDS_LUT && !Z1;
        auto true_end = [&](int j) -> int {
            if constexpr (ESC_LATE) {
                if (((unsigned)(cur.w[j] >> 32) & COV_PK_SPAN) == COV_PK_SPAN) {
                    const unsigned long long idx = t0 + (unsigned long long)tid * RPL + j;
                    return (I.end && idx < r1) ? I.end[idx] : sv[j];
                }
            }
            return ev[j];
        };
        if constexpr (PACKED) {
            bool esc = false;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const unsigned info = (unsigned)(cur.w[j] >> 32);
                sv[j] = (int)(unsigned)cur.w[j];
                ev[j] = sv[j] + (int)(info & COV_PK_SPAN);
                mq[j] = (info >> 24) & 63u;
                fl[j] = info >> 30;
                if (!ESC_LATE) esc = esc || (info & COV_PK_SPAN) == COV_PK_SPAN;
            }
            if (esc) {                                            // reads of >= 16 Mb: the end array has the truth
                const unsigned long long idx = t0 + (unsigned long long)tid * RPL;
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (((unsigned)(cur.w[j] >> 32) & COV_PK_SPAN) == COV_PK_SPAN) ev[j] = (I.end && idx + j < r1) ? I.end[idx + j] : sv[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                sv[j] = cur.s[j];
                ev[j] = cur.e[j];
                mq[j] = (cur.mq[j / 4] >> (8 * (j & 3))) & 0xffu;
                fl[j] = (cur.fl[j / 2] >> (16 * (j & 1))) & 0xffffu;
            }
        }

        // lane key K: first bin of the lane's first read (one division).  Any K is correct; sorted
        // input makes K non-decreasing across lanes.
        const int K = div(sv[0] < 0 ? 0 : sv[0]);
        const unsigned Kz = (unsigned)K * z;
        unsigned long long a0 = 0, a1 = 0, a2 = 0;
#ifdef COV_EXP_LOADONLY  // measurement variant: the loads alone (the 0.79 "load-only" ceiling quoted in DESIGN.md §3.1)
        for (int j = 0; j < RPL; j++) a0 += (unsigned)(sv[j] + ev[j]) + mq[j] + fl[j];
        if (a0 == 0x1234567ull) contribute(K, a0);
        cur = nxt;
        continue;
#endif
        unsigned slowmask = 0;
        constexpr bool TABLED = MODE == 0 && LDS_LUT && !Z1;      // MODE 0 with the pair tables in LDS (2 <= bin_size < 1024)
        if (TABLED) {
            // A read whose first bin is K or K+1 and whose last bin is at most one further is folded into the three registers a0,a1,a2
            // (bins K, K+1, K+2).  Its two quotients come from 16-byte table entries that already carry the routing: pairA[i] = (v, 0),
            // pairB[i] = (0, v); the first-bin entry adds (to K, to K+1), the last-bin entry one bin further (to K+1, to K+2); entry 0 is
            // (0, 0), so a read that does not take this path selects it and adds nothing — no predicate outlives its read (the
            // compare-and-select form of this loop kept eight reads' masks alive and spilled them with v_writelane).  Lanes whose bins
            // K..K+2 leave the window or touch the contig's last bin (other denominator, :66-69) send all their reads down the literal path.
            const ulonglong2 *pair = reinterpret_cast<const ulonglong2 *>(lutS + 2 * (z + 1));
            const unsigned ko = (unsigned)(K - base);
            const bool safe = ko < (unsigned)(WIN - 3) && K + 3 <= last_bin;
            const unsigned kw = safe ? ko : 0u;
            const unsigned z3 = 3u * z;
            const char *pairA = reinterpret_cast<const char *>(pair), *pairB = reinterpret_cast<const char *>(pair + (z + 1u));
            const int nz = -(int)z;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const unsigned rs = (unsigned)sv[j] - Kz;    // offsets from the start of bin K: first base, one past the last base
                const unsigned re1 = (unsigned)ev[j] - Kz;
                const unsigned len = re1 - rs;               // e - s (wraps to a huge value when e <= s)
                // __main__.py:231-235 / tiddit_signal.pyx:171-181.  Packed records keep duplicate | unmapped | mapq in the top byte of
                // the high word, so "no flag and mapq >= min_q" is ONE range test of that word: [min_q << 24, 64 << 24)
                bool cand;
                if constexpr (PACKED) cand = (unsigned)(cur.w[j] >> 32) - ((unsigned)P.min_q << 24) < ((64u - (unsigned)P.min_q) << 24);
                else cand = ((fl[j] & FMASK) == 0) & ((int)mq[j] >= P.min_q);
                const bool r1_ = rs >= z;                    // first bin = K + 1
                const unsigned rl = re1 - 1u;                                    // offset of the last base
                const unsigned dq = __umul24(rl, P.m15) >> P.k15;                // (e - 1 - K z) / z, exact for e - 1 - K z < 2^15
                const unsigned r = r1_ ? 1u : 0u;
                const bool fast = cand & safe & (rs < 2u * z) & (len - 1u < z3) & (re1 <= z3) & (dq - r <= 1u);
                nkept += fast ? 1u : 0u;
                slowmask |= (cand & !fast) ? (1u << j) : 0u;
                const bool multi = fast & (dq != r);
                const unsigned bf = min(len, (r1_ ? 2u * z : z) - rs);          // bases in the first bin (:55 / :61)
                const char *sect = r1_ ? pairB : pairA;                          // the (0, v) / (v, 0) table: one select of two bases
                const ulonglong2 X = *reinterpret_cast<const ulonglong2 *>(sect + ((fast ? bf : 0u) << 4));
                const unsigned bl = (unsigned)(__mul24((int)dq, nz) + (int)rl);  // bases in the last bin, counted one short as :63 does (one v_mad_i32_i24)
                const ulonglong2 Y = *reinterpret_cast<const ulonglong2 *>(sect + ((multi ? bl : 0u) << 4));
                a0 += X.x;
                a1 += X.y + Y.x;
                a2 += Y.y;
            }
            if (slowmask) {
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (slowmask & (1u << j)) {
                        const int s_ = sv[j], e_ = true_end(j);
                        if (s_ < 0 || e_ <= s_ || div(e_ - 1) > last_bin) bad = true;
                        else { nkept++; slow_read(s_, e_); }
                    }
            }
            // wavefront merge: inclusive prefix sums of the three registers; a run [a..b] of equal K sums to P[b] - P[a-1], so run-tail
            // lanes add +P[b] and run-head lanes add -P[a-1] (two's complement).  Only lanes on the window path carry sums, and a run
            // shares K, hence `safe`.
            wave_scan3_u64(a0, a1, a2);
            const int Kprev = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHR1, 0xf, 0xf, false);
            const int Knext = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHL1, 0xf, 0xf, false);
            const unsigned long long q0 = dpp_u64<DPP_WAVE_SHR1>(a0), q1 = dpp_u64<DPP_WAVE_SHR1>(a1), q2 = dpp_u64<DPP_WAVE_SHR1>(a2);
#ifndef COV_EXP_NOATOMIC
            if (safe && Knext != K) {  // run tail (lane 63 always: it reads ~K)
                if (a0) atomicAdd(&win[kw], a0);
                if (a1) atomicAdd(&win[kw + 1], a1);
                if (a2) atomicAdd(&win[kw + 2], a2);
            }
            if (safe && Kprev != K && lane != 0) {  // run head
                if (q0) atomicAdd(&win[kw], 0ull - q0);
                if (q1) atomicAdd(&win[kw + 1], 0ull - q1);
                if (q2) atomicAdd(&win[kw + 2], 0ull - q2);
            }
#endif
        } else if (MODE == 0) {
            // (bin_size 1, or >= 1024 with the tables in global memory) compare-and-select form of the same folding:
            // A read whose first bin is K or K+1 and whose last bin is at most one further is folded into the three
            // registers a0,a1,a2 (bins K, K+1, K+2) without branches; anything else (reads spanning more bins,
            // unsorted input) is flagged and replayed literally below.
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const int s = sv[j], e = ev[j];
                bool keep = !(fl[j] & FMASK) && (int)mq[j] >= P.min_q;   // __main__.py:231-235 / tiddit_signal.pyx:171-181
                const bool invalid = keep && (s < 0 || e <= s);
                const unsigned rs = (unsigned)s - Kz;        // offsets from the start of bin K
                const unsigned re = (unsigned)(e - 1) - Kz;
                const unsigned r = (rs >= z) ? 1u : 0u;      // first bin = K + r   (rs < 2z on the register path)
                const unsigned q = (re >= z ? 1u : 0u) + (re >= 2u * z ? 1u : 0u);  // last bin = K + q (re < 3z)
                const bool over = keep && !invalid && (K + (int)q > last_bin) && re < 3u * z;
                bad = bad || invalid || over;
                keep = keep && !invalid && !over;
                nkept += keep ? 1u : 0u;
                const bool fast = rs < 2u * z && re < 3u * z && q - r <= 1u;
                const bool ok = keep && fast;
                slowmask |= (keep && !fast) ? (1u << j) : 0u;
                const bool multi = q != r;
                // bases in the first bin (:55 single-bin / :61 multi-bin) and in the last bin (:63, one short)
                const unsigned bf = multi ? (r + 1u) * z - rs : (unsigned)(e - s);
                const unsigned bl = re - q * z + ((K + (int)q == last_bin) ? z + 1u : 0u);   // :66-69 picks the end-bin table
                const unsigned long long vF = lut(ok ? bf : 0u);                 // :57 / :62
                const unsigned long long vL = lut((ok && multi) ? bl : 0u);      // :66-69
                const bool r0_ = r == 0;
                a0 += r0_ ? vF : 0ull;
                a1 += r0_ ? vL : vF;
                a2 += r0_ ? 0ull : vL;
            }
            if (slowmask) {
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (slowmask & (1u << j)) {
                        if (div(ev[j] - 1) > last_bin) { bad = true; nkept--; }
                        else slow_read(sv[j], ev[j]);
                    }
            }
            // wavefront merge: inclusive prefix sums of the three registers; a run [a..b] of equal K sums to
            // P[b] - P[a-1], so run-tail lanes add +P[b] and run-head lanes add -P[a-1] (two's complement).
            // (cross-lane reads happen with every lane active: a masked-off DPP source lane returns nothing)
            wave_scan3_u64(a0, a1, a2);
            const int Kprev = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHR1, 0xf, 0xf, false);
            const int Knext = (int)__builtin_amdgcn_update_dpp((unsigned)~K, (unsigned)K, DPP_WAVE_SHL1, 0xf, 0xf, false);
            const unsigned long long q0 = dpp_u64<DPP_WAVE_SHR1>(a0), q1 = dpp_u64<DPP_WAVE_SHR1>(a1), q2 = dpp_u64<DPP_WAVE_SHR1>(a2);
            if (Knext != K) {  // run tail (lane 63 always: it reads ~K)
                if (a0) contribute(K, a0);
                if (a1) contribute(K + 1, a1);
                if (a2) contribute(K + 2, a2);
            }
            if (Kprev != K && lane != 0) {  // run head
                if (q0) contribute(K, 0ull - q0);
                if (q1) contribute(K + 1, 0ull - q1);
                if (q2) contribute(K + 2, 0ull - q2);
            }
        } else {
            const unsigned long long *tabA = lutS + 2 * (z + 1), *tabL = tabA + 2 * (z + 1);
            // a lane takes the window path when every bin its reads can reach there, K .. K + COV_DQMAX, lies inside the
            // window and before the contig's last bin (whose denominator differs, :66-69)
            const unsigned ko = (unsigned)(K - base);
            const bool safe = ko < (unsigned)(WIN - COV_DQMAX - 2) && K + COV_DQMAX + 2 <= last_bin;
            const unsigned kw = safe ? ko : 0u;
            unsigned d1 = 0, d2 = 0;                         // reads that put their +1 on bin K+1 / K+2
            unsigned long long vL[RPL];
            unsigned woff[RPL];
            // phase 1: table reads only (no LDS atomic in between, so nothing orders them behind one another)
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const unsigned rs = (unsigned)sv[j] - Kz;    // offsets from the start of bin K: first base, one past the last base
                const unsigned re1 = (unsigned)ev[j] - Kz;
                const unsigned len = re1 - rs;               // e - s (wraps to a huge value when e <= s)
                // __main__.py:231-235 / tiddit_signal.pyx:171-181.  Packed records keep duplicate | unmapped | mapq in the top byte of
                // the high word, so "no flag and mapq >= min_q" is ONE range test of that word: [min_q << 24, 64 << 24)
                bool cand;
                if constexpr (PACKED) cand = (unsigned)(cur.w[j] >> 32) - ((unsigned)P.min_q << 24) < ((64u - (unsigned)P.min_q) << 24);
                else cand = ((fl[j] & FMASK) == 0) & ((int)mq[j] >= P.min_q);
                // window path: first bin K or K+1, at most xmax bases past the start of bin K (so the 24-bit multiply below
                // is an exact division and the last bin is at most K + COV_DQMAX)
                const bool fast = cand & safe & (rs < 2u * z) & (len - 1u < P.xmax) & (re1 <= P.xmax);
                nkept += fast ? 1u : 0u;
                slowmask |= (cand & !fast) ? (1u << j) : 0u;
                const bool r1_ = rs >= z;                    // first bin = K + 1
                const unsigned dq = (__umul24(re1, P.m15) - P.m15) >> P.k15;     // (e - 1 - K z) / z: last bin = K + dq
                const bool multi = fast & (dq != (r1_ ? 1u : 0u));
                const unsigned bf = min(len, (r1_ ? 2u * z : z) - rs);          // bases in the first bin (:55 / :61)
                const unsigned long long vv = tabA[(fast ? bf : 0u) + (r1_ ? z + 1u : 0u)];   // (to bin K, to bin K+1)
                a0 += (unsigned)vv;
                a1 += (unsigned)(vv >> 32);
                d1 += (multi & !r1_) ? 1u : 0u;
                d2 += (multi & r1_) ? 1u : 0u;
                // last bin: re1 - dq z - 1 bases (:63, one short) with the -1 of the difference pair; masked reads add 0 to the spare word
                const unsigned bl1 = re1 - __umul24(dq, z);
                vL[j] = tabL[multi ? bl1 - 1u : z + 1u];
                woff[j] = multi ? kw + dq : (unsigned)WIN;
            }
#ifndef COV_EXP_NOATOMIC
            // phase 2: the atomics
#pragma unroll
            for (int j = 0; j < RPL; j++) atomicAdd(&win[woff[j]], vL[j]);
            atomicAdd(&win[kw], a0);
            atomicAdd(&win[kw + 1], a1 + ((unsigned long long)d1 << COV_DBIT));
            atomicAdd(&win[kw + 2], (unsigned long long)d2 << COV_DBIT);
#endif
            if (slowmask) {
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (slowmask & (1u << j)) {
                        const int s = sv[j], e = true_end(j);
                        if (s < 0 || e <= s || div(e - 1) > last_bin) bad = true;
                        else { nkept++; slow_read(s, e); }
                    }
            }
        }
        cur = nxt;
    }

    // kept-read count: one atomic per wave, spread over COV_KEPT_SLOTS cache lines (thousands of
    // same-address atomics serialise at ~12 ns each and would bound the whole launch)
    for (int d = 32; d > 0; d >>= 1) nkept += __shfl_down(nkept, d);
    nkept += nkept_s;
    if (lane == 0 && nkept)
        atomicAdd(P.kept + (size_t)((blockIdx.x * (COV_THREADS / 64) + (tid >> 6)) % COV_KEPT_SLOTS) * 16, (unsigned long long)nkept);
    if (bad) atomicOr(P.status, 1);

    spill();
}

__global__ void cov_finalize(const long long *__restrict__ acc, double *__restrict__ out, long long n, double inv_scale,
                             int *status) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    bool inexact = false;
    for (; i < n; i += stride) {
        const long long a = acc[i];
        if (a >= (1ll << 53) || a < 0) inexact = true;
        out[i] = (double)a * inv_scale;
    }
    if (inexact) atomicOr(status, 2);
}

// ------------------------------------------------------------------------------------------ host
struct tdt_cov {
    tdt_ctx *ctx = nullptr;
    int n_contigs = 0;
    int bin_size = 0;
    int S = 0;  // fixed-point fraction bits
    bool small_bins = false;  // cov_accumulate MODE 1
    unsigned xmax = 0, m15 = 0;
    int k15 = 0;
    unsigned magic = 0;
    int shift = 0;
    std::vector<int64_t> len, nbins, off;  // off: accumulator offset of each contig
    std::vector<int> end_bin_size;
    int64_t total_bins = 0;
    unsigned long long *d_acc = nullptr;
    unsigned long long *d_lut_main = nullptr;
    unsigned long long *d_lut_end = nullptr;  // [n_contigs][bin_size+1]
    int *d_status = nullptr;                  // [0] status bits; kept counters start 128 B further
    unsigned long long *d_kept = nullptr;     // COV_KEPT_SLOTS counters, 128 B apart
    int *d_nbins = nullptr;                   // bins of every contig (what cov_bin_record needs on the device)
    // staging slots for host pushes
    void *d_stage[2] = {nullptr, nullptr};
    void *h_stage[2] = {nullptr, nullptr};
    hipEvent_t slot_ev[2] = {nullptr, nullptr};
    int slot_used[2] = {0, 0};
    int next_slot = 0;
};

static bool build_lut(std::vector<unsigned long long> &lut, int bin_size, int den, int S) {
    const double scale = ldexp(1.0, S);
    for (int b = 0; b <= bin_size; b++) {
        // exactly what Cython emits: (float)bases / (float)den, widened to double
        volatile float q = (float)b / (float)den;
        double f = (double)q * scale;
        unsigned long long fx = (unsigned long long)f;
        if ((double)fx != f) return false;  // not representable at S bits: S too small
        lut[b] = fx;
    }
    return true;
}

extern "C" int tdt_cov_create(tdt_ctx *ctx, const int64_t *contig_len, int n_contigs, int bin_size, tdt_cov **out) {
    if (!ctx || !contig_len || !out || n_contigs <= 0) {
        tdt_set_error("tdt_cov_create: bad argument");
        return TDT_E_ARG;
    }
    *out = nullptr;
    if (bin_size <= 0 || bin_size > (1 << 24)) {
        tdt_set_error("tdt_cov_create: bin_size %d outside [1, 2^24] (reference: float32 bases are exact only there)",
                      bin_size);
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    tdt_cov *c = new tdt_cov();
    c->ctx = ctx;
    c->n_contigs = n_contigs;
    c->bin_size = bin_size;
    const int L = tdt_ceil_log2_u64((uint64_t)bin_size);
    c->S = 23 + L + 1;
    {
        // MODE 1 needs the float32 quotients to fit 32 bits (S <= 31: bin_size <= 128) and an exact 24-bit division
        const char *env = getenv("TIDDIT_COV_MODE");   // 0 / 1 force a kernel flavour (A/B measurements)
        const bool can = bin_size >= 2 && c->S <= 31;
        c->small_bins = can;
        if (env && env[0] == '0') c->small_bins = false;
        if (bin_size >= 2 && bin_size + 1 <= COV_LUT_LDS_MAX) {
            // floor(x / bin_size) == (x * m15) >> k15 for 0 <= x < 2^15 with a 24-bit multiply (the product stays below 2^31)
            c->k15 = 15 + L;
            c->m15 = (unsigned)(((1ull << c->k15) + (unsigned)bin_size - 1) / (unsigned)bin_size);
            const unsigned lim = (unsigned)COV_DQMAX * (unsigned)bin_size;
            c->xmax = lim < 32768u ? lim : 32768u;
            for (unsigned x = 0; x < 32768u; x++)
                if (((x * c->m15) >> c->k15) != x / (unsigned)bin_size) {
                    tdt_set_error("internal: 24-bit division self-check failed for d=%d x=%u", bin_size, x);
                    delete c;
                    return TDT_E_ARG;
                }
        }
    }
    if (L == 0) {
        c->shift = -1;
        c->magic = 0;
    } else {
        // m = ceil(2^(31+L) / d): floor(x/d) == (x*m) >> (31+L) for all 0 <= x < 2^31
        const unsigned __int128 num = (unsigned __int128)1 << (31 + L);
        const unsigned __int128 m = (num + (unsigned)bin_size - 1) / (unsigned)bin_size;
        c->magic = (unsigned)m;
        c->shift = L - 1;
        const int probes[] = {0, 1, bin_size - 1, bin_size, bin_size + 1, 2 * bin_size - 1, 2 * bin_size, 1000003,
                              0x3fffffff, 0x7ffffffe, 0x7fffffff};
        for (int x : probes) {
            if (x < 0) continue;
            const int q = (int)(((unsigned long long)(unsigned)x * c->magic) >> 32 >> c->shift);
            if (q != x / bin_size) {
                tdt_set_error("internal: magic division self-check failed for d=%d x=%d", bin_size, x);
                delete c;
                return TDT_E_ARG;
            }
        }
    }
    c->len.assign(contig_len, contig_len + n_contigs);
    c->nbins.resize(n_contigs);
    c->off.resize(n_contigs);
    c->end_bin_size.resize(n_contigs);
    int64_t total = 0;
    for (int i = 0; i < n_contigs; i++) {
        const int64_t LN = contig_len[i];
        if (LN < 0 || LN > 0x7fffffffll) {
            tdt_set_error("tdt_cov_create: contig %d length %lld outside [0, 2^31)", i, (long long)LN);
            delete c;
            return TDT_E_ARG;
        }
        const int64_t bins = (LN + bin_size - 1) / bin_size;  // == int(ceil(LN/float(bin_size))) for LN < 2^53
        c->nbins[i] = bins;
        c->end_bin_size[i] = (int)(LN - (bins - 1) * bin_size);
        c->off[i] = total;
        total += bins;
        // keep every contig's accumulators 16-byte aligned
        total = (total + 1) & ~1ll;
    }
    c->total_bins = total;
    const size_t lut_n = (size_t)bin_size + 1;
    std::vector<unsigned long long> lut(lut_n), lute((size_t)n_contigs * lut_n);
    if (!build_lut(lut, bin_size, bin_size, c->S)) {
        tdt_set_error("internal: fixed-point scale 2^-%d cannot represent float32(b)/float32(%d)", c->S, bin_size);
        delete c;
        return TDT_E_ARG;
    }
    for (int i = 0; i < n_contigs; i++) {
        std::vector<unsigned long long> t(lut_n);
        const int ebs = c->end_bin_size[i] > 0 ? c->end_bin_size[i] : bin_size;
        // b/ebs >= b/bin_size, so the quotient never needs more fraction bits than the main table.
        // Entries with b >= ebs are only read for reads overhanging the contig end inside its last
        // bin (the reference then really divides by end_bin_size, :69); they are exact too.
        const double scale = ldexp(1.0, c->S);
        for (size_t b = 0; b < lut_n; b++) {
            volatile float q = (float)b / (float)ebs;
            double f = (double)q * scale;
            unsigned long long fx = (f < 1.8e19) ? (unsigned long long)f : 0ull;
            t[b] = ((double)fx == f) ? fx : 0ull;
        }
        memcpy(&lute[(size_t)i * lut_n], t.data(), lut_n * sizeof(unsigned long long));
    }
    hipError_t e;
    auto fail = [&](const char *what) {
        tdt_set_error("tdt_cov_create: %s failed: %s", what, hipGetErrorString(e));
        tdt_cov_destroy(c);
        return TDT_E_HIP;
    };
    if ((e = tdt_dev_malloc((void **)&c->d_acc, (size_t)(total > 0 ? total : 1) * 8)) != hipSuccess) return fail("hipMalloc(acc)");
    if ((e = tdt_dev_malloc((void **)&c->d_lut_main, lut_n * 8)) != hipSuccess) return fail("hipMalloc(lut)");
    if ((e = tdt_dev_malloc((void **)&c->d_lut_end, (size_t)n_contigs * lut_n * 8)) != hipSuccess) return fail("hipMalloc(lut_end)");
    if ((e = tdt_dev_malloc((void **)&c->d_status, COV_STATUS_BYTES)) != hipSuccess) return fail("hipMalloc(status)");
    {
        std::vector<int> nb32(c->nbins.begin(), c->nbins.end());
        if ((e = tdt_dev_malloc((void **)&c->d_nbins, (size_t)n_contigs * 4)) != hipSuccess) return fail("hipMalloc(nbins)");
        if ((e = hipMemcpy(c->d_nbins, nb32.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(nbins)");
    }
    c->d_kept = (unsigned long long *)((char *)c->d_status + 128);
    if ((e = hipMemcpy(c->d_lut_main, lut.data(), lut_n * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(lut)");
    if ((e = hipMemcpy(c->d_lut_end, lute.data(), (size_t)n_contigs * lut_n * 8, hipMemcpyHostToDevice)) != hipSuccess)
        return fail("hipMemcpy(lut_end)");
    if ((e = hipMemset(c->d_acc, 0, (size_t)(total > 0 ? total : 1) * 8)) != hipSuccess) return fail("hipMemset(acc)");
    if ((e = hipMemset(c->d_status, 0, COV_STATUS_BYTES)) != hipSuccess) return fail("hipMemset(status)");
    for (int i = 0; i < 2; i++)
        if ((e = hipEventCreateWithFlags(&c->slot_ev[i], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate");
    *out = c;
    return TDT_OK;
}

extern "C" void tdt_cov_destroy(tdt_cov *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->d_lut_main) (void)hipFree(c->d_lut_main);
    if (c->d_lut_end) (void)hipFree(c->d_lut_end);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_nbins) (void)hipFree(c->d_nbins);
    for (int i = 0; i < 2; i++) {
        if (c->d_stage[i]) (void)hipFree(c->d_stage[i]);
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->slot_ev[i]) (void)hipEventDestroy(c->slot_ev[i]);
    }
    delete c;
}

extern "C" int tdt_cov_nbins(tdt_cov *c, int tid, int64_t *nbins, int *end_bin_size) {
    if (!c || tid < 0 || tid >= c->n_contigs) {
        tdt_set_error("tdt_cov_nbins: bad contig id");
        return TDT_E_ARG;
    }
    if (nbins) *nbins = c->nbins[tid];
    if (end_bin_size) *end_bin_size = c->end_bin_size[tid];
    return TDT_OK;
}

extern "C" int tdt_cov_scale_bits(tdt_cov *c) { return c ? c->S : TDT_E_ARG; }

extern "C" int tdt_cov_reset(tdt_cov *c) {
    if (!c) return TDT_E_ARG;
    TDT_HIP(hipSetDevice(c->ctx->device));
    TDT_HIP(hipMemsetAsync(c->d_acc, 0, (size_t)(c->total_bins > 0 ? c->total_bins : 1) * 8, c->ctx->stream));
    TDT_HIP(hipMemsetAsync(c->d_status, 0, COV_STATUS_BYTES, c->ctx->stream));
    return TDT_OK;
}

static CovItem cov_item(tdt_cov *c, int tid, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq,
                        const uint16_t *d_flag, size_t n) {
    CovItem it;
    it.start = d_start;
    it.end = d_end;
    it.mapq = d_mapq;
    it.flag = d_flag;
    it.packed = nullptr;
    it.n = n;
    it.acc = c->d_acc + c->off[tid];
    it.lut_end = c->d_lut_end + (size_t)tid * ((size_t)c->bin_size + 1);
    it.nbins = (int)c->nbins[tid];
    it.aligned = (((uintptr_t)d_start | (uintptr_t)d_end) & 15) == 0 && ((uintptr_t)d_mapq & (COV_RPL - 1)) == 0 &&
                 ((uintptr_t)d_flag & (2 * COV_RPL - 1)) == 0;
    it.first_block = 0;
    it.binned = 0;
    return it;
}

static CovItem cov_item_packed(tdt_cov *c, int tid, const unsigned long long *d_packed, const int32_t *d_end, size_t n) {
    CovItem it = cov_item(c, tid, nullptr, d_end, nullptr, nullptr, n);
    it.packed = d_packed;
    it.aligned = ((uintptr_t)d_packed & 15) == 0;
    return it;
}

static int cov_launch_items(tdt_cov *c, const CovItem &single, const CovItem *d_items, int n_items, unsigned grid, int min_q) {
    CovParams P;
    P.it = single;
    P.items = d_items;
    P.n_items = n_items;
    P.bin_size = c->bin_size;
    P.magic = c->magic;
    P.shift = c->shift;
    P.min_q = min_q;
    P.lut_main = c->d_lut_main;
    P.one = 1ull << c->S;
    P.status = c->d_status;
    P.kept = c->d_kept;
    const bool small = c->small_bins;
    P.margin = small ? COV_DQMAX + 8 : 640 / c->bin_size + 4;
    P.xmax = c->xmax;
    P.m15 = c->m15;
    P.k15 = c->k15;
    const bool lds_lut = c->bin_size + 1 <= COV_LUT_LDS_MAX;
    const size_t lds = 96 + (small ? 128 : 0) + ((size_t)(small ? COV_WIN1 + COV_SPARES1 : COV_WIN + 2)) * 8 + (lds_lut ? 2 * ((size_t)c->bin_size + 1) * 8 : 0) +
                       (small ? (3 * ((size_t)c->bin_size + 1) + 1) * 8 : 0) + ((!small && lds_lut && c->shift >= 0) ? 4 * ((size_t)c->bin_size + 1) * 8 : 0);
    // small bins: few reads share a bin, a read covers several -> difference-pair kernel
    const bool packed = single.packed != nullptr;
    const bool binned = packed && single.binned;              // (one layout per launch: the push entry points build their items alike)
    if (binned && small)
        hipLaunchKernelGGL((cov_accumulate<true, 1, false, COV_RPL1, 2>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (binned)
        hipLaunchKernelGGL((cov_accumulate<true, 0, false, COV_RPL, 2>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (small && packed)
        hipLaunchKernelGGL((cov_accumulate<true, 1, false, COV_RPL1, 1>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (small)
        hipLaunchKernelGGL((cov_accumulate<true, 1, false, COV_RPL1, 0>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (lds_lut && c->shift >= 0 && packed)
        hipLaunchKernelGGL((cov_accumulate<true, 0, false, COV_RPL, 1>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (lds_lut && c->shift >= 0)
        hipLaunchKernelGGL((cov_accumulate<true, 0, false, COV_RPL, 0>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (lds_lut && packed)
        hipLaunchKernelGGL((cov_accumulate<true, 0, true, COV_RPL, 1>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (lds_lut)
        hipLaunchKernelGGL((cov_accumulate<true, 0, true, COV_RPL, 0>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else if (packed)
        hipLaunchKernelGGL((cov_accumulate<false, 0, false, COV_RPL, 1>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    else
        hipLaunchKernelGGL((cov_accumulate<false, 0, false, COV_RPL, 0>), dim3(grid), dim3(COV_THREADS), lds, c->ctx->stream, P);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

static int cov_launch(tdt_cov *c, int tid, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq,
                      const uint16_t *d_flag, size_t n, int min_q) {
    if (n == 0 || c->nbins[tid] == 0) return TDT_OK;
    const unsigned grid = (unsigned)((n + COV_READS_PER_BLOCK - 1) / COV_READS_PER_BLOCK);
    return cov_launch_items(c, cov_item(c, tid, d_start, d_end, d_mapq, d_flag, n), nullptr, 0, grid, min_q);
}

extern "C" int tdt_cov_push_device_multi(tdt_cov *c, int n_items, const int *tids, const int32_t *const *d_start,
                                         const int32_t *const *d_end, const uint8_t *const *d_mapq,
                                         const uint16_t *const *d_flag, const size_t *n, int min_q) {
    if (!c || n_items < 0 || (n_items && (!tids || !d_start || !d_end || !d_mapq || !d_flag || !n))) {
        tdt_set_error("tdt_cov_push_device_multi: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    std::vector<CovItem> items;
    unsigned long long blocks = 0;
    for (int i = 0; i < n_items; i++) {
        if (tids[i] < 0 || tids[i] >= c->n_contigs || (n[i] && (!d_start[i] || !d_end[i] || !d_mapq[i] || !d_flag[i]))) {
            tdt_set_error("tdt_cov_push_device_multi: bad item %d", i);
            return TDT_E_ARG;
        }
        if (n[i] == 0 || c->nbins[tids[i]] == 0) continue;
        CovItem it = cov_item(c, tids[i], d_start[i], d_end[i], d_mapq[i], d_flag[i], n[i]);
        it.first_block = (unsigned)blocks;
        blocks += (n[i] + COV_READS_PER_BLOCK - 1) / COV_READS_PER_BLOCK;
        items.push_back(it);
    }
    if (items.empty()) return TDT_OK;
    if (blocks >= 0x7fffffffull) {
        tdt_set_error("tdt_cov_push_device_multi: too many reads for one launch");
        return TDT_E_ARG;
    }
    void *d_items = nullptr;
    int rc = tdt_scratch(c->ctx, 7, items.size() * sizeof(CovItem), &d_items);
    if (rc) return rc;
    // pageable source: the runtime stages it before returning, so `items` may go out of scope
    TDT_HIP(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(CovItem), hipMemcpyHostToDevice, c->ctx->stream));
    return cov_launch_items(c, items[0], (const CovItem *)d_items, (int)items.size(), (unsigned)blocks, min_q);
}

extern "C" int tdt_cov_pack_device(tdt_ctx *ctx, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq, const uint16_t *d_flag,
                                   size_t n, uint64_t *d_packed) {
    if (!ctx || (n && (!d_start || !d_end || !d_mapq || !d_flag || !d_packed))) {
        tdt_set_error("tdt_cov_pack_device: bad argument");
        return TDT_E_ARG;
    }
    if (!n) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(cov_pack, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_start, d_end, d_mapq, d_flag, (unsigned long long)n,
                       (unsigned long long *)d_packed);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_cov_push_packed_device_multi(tdt_cov *c, int n_items, const int *tids, const uint64_t *const *d_packed,
                                                const int32_t *const *d_end, const size_t *n, int min_q) {
    if (!c || n_items < 0 || (n_items && (!tids || !d_packed || !n))) {
        tdt_set_error("tdt_cov_push_packed_device_multi: bad argument");
        return TDT_E_ARG;
    }
    if (min_q > 63) {
        tdt_set_error("tdt_cov_push_packed_device_multi: packed records keep min(mapq, 63); min_q %d needs the unpacked entry point", min_q);
        return TDT_E_UNSUPPORTED;
    }
    if (min_q < 0) min_q = 0;                 // every mapq passes either way; the kernel's range test wants a field value
    TDT_HIP(hipSetDevice(c->ctx->device));
    std::vector<CovItem> items;
    unsigned long long blocks = 0;
    for (int i = 0; i < n_items; i++) {
        if (tids[i] < 0 || tids[i] >= c->n_contigs || (n[i] && !d_packed[i])) {
            tdt_set_error("tdt_cov_push_packed_device_multi: bad item %d", i);
            return TDT_E_ARG;
        }
        if (n[i] == 0 || c->nbins[tids[i]] == 0) continue;
        CovItem it = cov_item_packed(c, tids[i], (const unsigned long long *)d_packed[i], d_end ? d_end[i] : nullptr, n[i]);
        it.first_block = (unsigned)blocks;
        blocks += (n[i] + COV_READS_PER_BLOCK - 1) / COV_READS_PER_BLOCK;
        items.push_back(it);
    }
    if (items.empty()) return TDT_OK;
    if (blocks >= 0x7fffffffull) {
        tdt_set_error("tdt_cov_push_packed_device_multi: too many reads for one launch");
        return TDT_E_ARG;
    }
    void *d_items = nullptr;
    int rc = tdt_scratch(c->ctx, 7, items.size() * sizeof(CovItem), &d_items);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(CovItem), hipMemcpyHostToDevice, c->ctx->stream));
    return cov_launch_items(c, items[0], (const CovItem *)d_items, (int)items.size(), (unsigned)blocks, min_q);
}

// ---- binned records: cov_bin_record for THIS histogram's bin size (2 <= bin_size < 1024)
int tdt_cov_bin_spec(tdt_cov *c, CovBinSpec *out) {          // (C++ linkage: the ingest kernel writes such records itself, tdt_ingest.hip)
    *out = CovBinSpec();
    if (!c) return TDT_E_ARG;
    if (c->bin_size < 2 || c->bin_size + 1 > COV_LUT_LDS_MAX) return TDT_OK;       // z stays 0: no binned records for this bin size
    out->z = (unsigned)c->bin_size;
    out->magic = c->magic;
    out->shift = c->shift;
    out->mode1 = c->small_bins ? 1 : 0;
    out->d_nbins = c->d_nbins;
    out->n_contigs = c->n_contigs;
    return TDT_OK;
}

extern "C" int tdt_cov_pack_binned_device(tdt_cov *c, int tid, const int32_t *d_start, const int32_t *d_end, const uint8_t *d_mapq,
                                          const uint16_t *d_flag, size_t n, uint64_t *d_binned) {
    if (!c || tid < 0 || tid >= c->n_contigs || (n && (!d_start || !d_end || !d_mapq || !d_flag || !d_binned))) {
        tdt_set_error("tdt_cov_pack_binned_device: bad argument");
        return TDT_E_ARG;
    }
    CovBinSpec sp;
    tdt_cov_bin_spec(c, &sp);
    if (!sp.z) {
        tdt_set_error("tdt_cov_pack_binned_device: binned records exist for 2 <= bin_size < 1024 (this histogram: %d)", c->bin_size);
        return TDT_E_UNSUPPORTED;
    }
    if (!n) return TDT_OK;
    TDT_HIP(hipSetDevice(c->ctx->device));
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(cov_pack_binned, dim3((unsigned)blocks), dim3(256), 0, c->ctx->stream, d_start, d_end, d_mapq, d_flag, (unsigned long long)n,
                       (int)c->nbins[tid], sp.z, sp.magic, sp.shift, sp.mode1, (unsigned long long *)d_binned);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_cov_push_binned_device_multi(tdt_cov *c, int n_items, const int *tids, const uint64_t *const *d_binned,
                                                const int32_t *const *d_start, const int32_t *const *d_end, const size_t *n, int min_q) {
    if (!c || n_items < 0 || (n_items && (!tids || !d_binned || !d_start || !d_end || !n))) {
        tdt_set_error("tdt_cov_push_binned_device_multi: bad argument");
        return TDT_E_ARG;
    }
    if (c->bin_size < 2 || c->bin_size + 1 > COV_LUT_LDS_MAX) {
        tdt_set_error("tdt_cov_push_binned_device_multi: binned records exist for 2 <= bin_size < 1024 (this histogram: %d)", c->bin_size);
        return TDT_E_UNSUPPORTED;
    }
    if (min_q > 63) {
        tdt_set_error("tdt_cov_push_binned_device_multi: records keep min(mapq, 63); min_q %d needs the unpacked entry point", min_q);
        return TDT_E_UNSUPPORTED;
    }
    if (min_q < 0) min_q = 0;
    TDT_HIP(hipSetDevice(c->ctx->device));
    std::vector<CovItem> items;
    unsigned long long blocks = 0;
    for (int i = 0; i < n_items; i++) {
        if (tids[i] < 0 || tids[i] >= c->n_contigs || (n[i] && (!d_binned[i] || !d_start[i] || !d_end[i]))) {
            tdt_set_error("tdt_cov_push_binned_device_multi: bad item %d (the literal path needs the start and end arrays)", i);
            return TDT_E_ARG;
        }
        if (n[i] == 0 || c->nbins[tids[i]] == 0) continue;
        CovItem it = cov_item_packed(c, tids[i], (const unsigned long long *)d_binned[i], d_end[i], n[i]);
        it.start = d_start[i];
        it.binned = 1;
        it.first_block = (unsigned)blocks;
        blocks += (n[i] + COV_READS_PER_BLOCK - 1) / COV_READS_PER_BLOCK;
        items.push_back(it);
    }
    if (items.empty()) return TDT_OK;
    if (blocks >= 0x7fffffffull) {
        tdt_set_error("tdt_cov_push_binned_device_multi: too many reads for one launch");
        return TDT_E_ARG;
    }
    void *d_items = nullptr;
    int rc = tdt_scratch(c->ctx, 7, items.size() * sizeof(CovItem), &d_items);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(CovItem), hipMemcpyHostToDevice, c->ctx->stream));
    return cov_launch_items(c, items[0], (const CovItem *)d_items, (int)items.size(), (unsigned)blocks, min_q);
}

extern "C" int tdt_cov_push_device(tdt_cov *c, int tid, const int32_t *d_start, const int32_t *d_end,
                                   const uint8_t *d_mapq, const uint16_t *d_flag, size_t n, int min_q) {
    if (!c || tid < 0 || tid >= c->n_contigs || (n && (!d_start || !d_end || !d_mapq || !d_flag))) {
        tdt_set_error("tdt_cov_push_device: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    return cov_launch(c, tid, d_start, d_end, d_mapq, d_flag, n, min_q);
}

extern "C" int tdt_cov_push(tdt_cov *c, int tid, const int32_t *start, const int32_t *end, const uint8_t *mapq,
                            const uint16_t *flag, size_t n, int min_q) {
    if (!c || tid < 0 || tid >= c->n_contigs || (n && (!start || !end || !mapq || !flag))) {
        tdt_set_error("tdt_cov_push: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    const size_t chunk = COV_PUSH_CHUNK;
    const size_t slot_bytes = chunk * 12;  // 4+4+2+1 B/read, each array 16-byte aligned (chunk % 16 == 0) -> 11 B + slack
    for (size_t o = 0; o < n; o += chunk) {
        const size_t m = n - o < chunk ? n - o : chunk;
        const int s = c->next_slot;
        c->next_slot ^= 1;
        if (!c->h_stage[s]) {
            TDT_HIP(hipHostMalloc(&c->h_stage[s], slot_bytes, hipHostMallocDefault));
            TDT_HIP(tdt_dev_malloc(&c->d_stage[s], slot_bytes));
        }
        if (c->slot_used[s]) TDT_HIP(hipEventSynchronize(c->slot_ev[s]));
        char *h = (char *)c->h_stage[s];
        char *d = (char *)c->d_stage[s];
        const size_t o_end = chunk * 4, o_flag = chunk * 8, o_mapq = chunk * 10;
        if (m >= (1u << 18) && tdt_host_thread_count() >= 4) {   // the staging copy bounds this path: one thread per array
            std::thread t1([&] { memcpy(h, start + o, m * 4); });   // (finer slices over 8+ threads measured slower: 18 vs 25 GB/s)
            std::thread t2([&] { memcpy(h + o_end, end + o, m * 4); });
            std::thread t3([&] { memcpy(h + o_flag, flag + o, m * 2); });
            memcpy(h + o_mapq, mapq + o, m);
            t1.join();
            t2.join();
            t3.join();
        } else {
            memcpy(h, start + o, m * 4);
            memcpy(h + o_end, end + o, m * 4);
            memcpy(h + o_flag, flag + o, m * 2);
            memcpy(h + o_mapq, mapq + o, m);
        }
        if (m == chunk) {
            TDT_HIP(hipMemcpyAsync(d, h, chunk * 11, hipMemcpyHostToDevice, c->ctx->stream));
        } else {
            TDT_HIP(hipMemcpyAsync(d, h, m * 4, hipMemcpyHostToDevice, c->ctx->stream));
            TDT_HIP(hipMemcpyAsync(d + o_end, h + o_end, m * 4, hipMemcpyHostToDevice, c->ctx->stream));
            TDT_HIP(hipMemcpyAsync(d + o_flag, h + o_flag, m * 2, hipMemcpyHostToDevice, c->ctx->stream));
            TDT_HIP(hipMemcpyAsync(d + o_mapq, h + o_mapq, m, hipMemcpyHostToDevice, c->ctx->stream));
        }
        int rc = cov_launch(c, tid, (const int32_t *)d, (const int32_t *)(d + o_end), (const uint8_t *)(d + o_mapq),
                            (const uint16_t *)(d + o_flag), m, min_q);
        if (rc) return rc;
        TDT_HIP(hipEventRecord(c->slot_ev[s], c->ctx->stream));
        c->slot_used[s] = 1;
    }
    return TDT_OK;
}

static int cov_status(tdt_cov *c) {
    int st[4];
    TDT_HIP(hipMemcpyAsync(st, c->d_status, 16, hipMemcpyDeviceToHost, c->ctx->stream));
    TDT_HIP(hipStreamSynchronize(c->ctx->stream));
    if (st[0] & 1) {
        tdt_set_error("coverage: a read maps outside its contig's bins (reference raises IndexError) or has end <= start");
        return TDT_E_RANGE;
    }
    if (st[0] & 2) {
        tdt_set_error("coverage: a bin exceeded 2^53 fixed-point units; the float64 sum is no longer exact");
        return TDT_E_INEXACT;
    }
    return TDT_OK;
}

extern "C" int tdt_cov_finish_device(tdt_cov *c, int tid, double *d_out) {
    if (!c || tid < 0 || tid >= c->n_contigs || (!d_out && c->nbins[tid])) {
        tdt_set_error("tdt_cov_finish_device: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    const long long nb = c->nbins[tid];
    if (nb) {
        const int threads = 256;
        long long blocks = (nb + threads - 1) / threads;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(cov_finalize, dim3((unsigned)blocks), dim3(threads), 0, c->ctx->stream,
                           (const long long *)(c->d_acc + c->off[tid]), d_out, nb, ldexp(1.0, -c->S), c->d_status);
        TDT_CHECK_LAUNCH();
    }
    return TDT_OK;
}

extern "C" int tdt_cov_total_bins(tdt_cov *c, int64_t *total) {
    if (!c || !total) return TDT_E_ARG;
    *total = c->total_bins;
    return TDT_OK;
}

extern "C" int tdt_cov_offset(tdt_cov *c, int tid, int64_t *off) {
    if (!c || !off || tid < 0 || tid >= c->n_contigs) return TDT_E_ARG;
    *off = c->off[tid];
    return TDT_OK;
}

extern "C" int tdt_cov_finish_all_device(tdt_cov *c, double *d_out) {
    if (!c || !d_out) {
        tdt_set_error("tdt_cov_finish_all_device: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    const long long nb = c->total_bins;
    if (nb) {
        const int threads = 256;
        long long blocks = (nb + threads - 1) / threads;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(cov_finalize, dim3((unsigned)blocks), dim3(threads), 0, c->ctx->stream, (const long long *)c->d_acc, d_out, nb,
                           ldexp(1.0, -c->S), c->d_status);
        TDT_CHECK_LAUNCH();
    }
    return TDT_OK;
}

extern "C" int tdt_cov_finish(tdt_cov *c, int tid, double *out) {
    if (!c || tid < 0 || tid >= c->n_contigs || (!out && c->nbins[tid])) {
        tdt_set_error("tdt_cov_finish: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    const size_t nb = (size_t)c->nbins[tid];
    void *d_out = nullptr;
    int rc = tdt_scratch(c->ctx, 0, (nb ? nb : 1) * 8, &d_out);
    if (rc) return rc;
    rc = tdt_cov_finish_device(c, tid, (double *)d_out);
    if (rc) return rc;
    if (nb) TDT_HIP(hipMemcpyAsync(out, d_out, nb * 8, hipMemcpyDeviceToHost, c->ctx->stream));
    return cov_status(c);
}

// every contig's bins at once, contig t at out + tdt_cov_offset(t): one finalize launch, one copy, one wait — a header with thousands
// of contigs (GRCh38 with its alt / decoy / HLA contigs) paid a launch + copy + wait per contig through tdt_cov_finish
extern "C" int tdt_cov_finish_all(tdt_cov *c, double *out) {
    if (!c || (!out && c->total_bins)) {
        tdt_set_error("tdt_cov_finish_all: bad argument");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->ctx->device));
    const size_t nb = (size_t)c->total_bins;
    if (!nb) return cov_status(c);
    void *d_out = nullptr;
    int rc = tdt_scratch(c->ctx, 0, nb * 8, &d_out);
    if (rc) return rc;
    rc = tdt_cov_finish_all_device(c, (double *)d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(out, d_out, nb * 8, hipMemcpyDeviceToHost, c->ctx->stream));
    return cov_status(c);
}

extern "C" int tdt_cov_kept(tdt_cov *c, int64_t *kept) {
    if (!c || !kept) return TDT_E_ARG;
    TDT_HIP(hipSetDevice(c->ctx->device));
    std::vector<unsigned long long> h((size_t)COV_KEPT_SLOTS * 16);
    TDT_HIP(hipMemcpyAsync(h.data(), c->d_kept, h.size() * 8, hipMemcpyDeviceToHost, c->ctx->stream));
    TDT_HIP(hipStreamSynchronize(c->ctx->stream));
    unsigned long long k = 0;
    for (int i = 0; i < COV_KEPT_SLOTS; i++) k += h[(size_t)i * 16];
    *kept = (int64_t)k;
    return TDT_OK;
}
