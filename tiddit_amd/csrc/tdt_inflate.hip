// BGZF inflate on the MI355X: one wavefront per BGZF block (RFC 1951 DEFLATE, <= 64 KiB of output each).
//
// The reference leaves decompression to pysam/htslib worker threads on the host (tiddit_signal.pyx:159,
// __main__.py:224); a GPU node has few host cores per device, so here the compressed blocks go over PCIe as they are
// and are inflated in HBM, next to the record decode and the coverage kernel that consume them.
//
// Per wavefront (all control flow is wave-uniform, state lives in SGPRs):
//   * input: the compressed stream is held in registers — lane l keeps dword l of the current and of the next 256-byte
//     window (two coalesced loads, the next window is in flight while the current one is consumed) and the bit buffer is
//     refilled with v_readlane; no per-symbol memory access on the input side;
//   * Huffman tables: per DEFLATE block, built by the whole wave in LDS from the code lengths (ballot counting sort ->
//     canonical codes -> bit-reversed LUT, 10 bits literal/length, 9 bits distance, canonical walk for longer codes);
//   * output: literals are collected one per lane (64 per coalesced store); a match is copied by all lanes at once
//     (out[p+i] = out[p-D+(i mod D)], every source byte precedes p); HBM writes of a wave are ordered, so a later match
//     reads what earlier stores of the same wave wrote;
//   * every loop is bounded by ISIZE / the compressed length, a malformed stream sets the block's status and stops.
// A second kernel checks each block's CRC32 (lane-parallel table CRC + GF(2) combine).
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_inflate;
const char *const tdt_variant_inflate = ""
#ifdef BZ_STATS
    " BZ_STATS"
#endif
    ;

#include "tdt_common.h"

#define BZ_TB_LL 10
#define BZ_TB_D 9
#define BZ_WAVES 4
// LDS bytes per wave: lens 320 | lut_ll 2048 | lut_d 1024 | sorted_ll 576 | sorted_d 64 | meta_ll 96 | meta_d 96
#define BZ_OFF_LUTLL 320
#define BZ_OFF_LUTD (BZ_OFF_LUTLL + 2048)
#define BZ_OFF_SORTLL (BZ_OFF_LUTD + 1024)
#define BZ_OFF_SORTD (BZ_OFF_SORTLL + 576)
#define BZ_OFF_METALL (BZ_OFF_SORTD + 64)
#define BZ_OFF_METAD (BZ_OFF_METALL + 96)
#define BZ_LDS (BZ_OFF_METAD + 96 + 8)

enum { BZ_OK = 0, BZ_E_BTYPE = 1, BZ_E_STORED = 2, BZ_E_TABLE = 3, BZ_E_SYMBOL = 4, BZ_E_DIST = 5, BZ_E_OVERRUN = 6, BZ_E_INPUT = 7,
       BZ_E_SIZE = 8, BZ_E_CRC = 9 };

typedef unsigned long long u64;

__device__ __forceinline__ unsigned bz_rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

#ifdef BZ_STATS
__device__ unsigned long long bz_stats[64];   // [0] literals [1] matches [2] match bytes [3] deflate blocks, [8+k] matches with dist < 2^k, [32+k] len < 2^k
// BZ_STATS measurement builds only (tools/inflate_stats.py); not part of the shipped ABI
extern "C" int tdt_debug_bz_stats(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long z[64] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(bz_stats), z, sizeof z) == hipSuccess ? 0 : -2;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bz_stats), 64 * 8) == hipSuccess ? 0 : -2;
}
#endif

__constant__ unsigned char bz_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Canonical Huffman tables for `n` code lengths in LDS.  lut: (symbol << 4 | length) indexed by the next `tb` stream bits
// (0 = longer code or unused), sorted: symbols ordered by (length, symbol), meta: first code / count / offset per length.
// Lane k carries the running state of code length k (counts, first code, output cursor), so nothing is unrolled into
// registers.  Returns false when the lengths over-subscribe the code space.
__device__ __forceinline__ bool bz_build(const unsigned char *lens, int n, int tb, unsigned short *lut, unsigned short *sorted,
                                         unsigned short *meta, int lane) {
    for (int i = lane; i < (1 << tb) / 2; i += 64) ((unsigned *)lut)[i] = 0;
    unsigned cntv = 0;                                           // lane k: number of codes of length k
#pragma nounroll
    for (int c = 0; c < n; c += 64) {
        const int s = c + lane;
        const unsigned l = s < n ? lens[s] : 0;
#pragma nounroll
        for (int k = 1; k < 16; k++) {
            const unsigned m = (unsigned)__popcll(__ballot(l == (unsigned)k));
            cntv += lane == k ? m : 0;
        }
    }
    int left = 1;
    bool over = false;
    unsigned code = 0, o = 0, prev = 0, firstv = 0, offv = 0;
#pragma nounroll
    for (int k = 1; k < 16; k++) {
        const unsigned ck = (unsigned)__builtin_amdgcn_readlane((int)cntv, k);
        left = (left << 1) - (int)ck;
        over = over || left < 0;
        code = (code + prev) << 1;
        firstv = lane == k ? code : firstv;
        offv = lane == k ? o : offv;
        o += ck;
        prev = ck;
    }
    if (over) return false;
    if (lane < 16) {
        meta[lane] = (unsigned short)firstv;
        meta[16 + lane] = (unsigned short)cntv;
        meta[32 + lane] = (unsigned short)offv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned runv = offv;                                        // lane k: next slot of `sorted` for length k
    const u64 below = (1ull << lane) - 1ull;
#pragma nounroll
    for (int c = 0; c < n; c += 64) {
        const int s = c + lane;
        const unsigned l = s < n ? lens[s] : 0;
        unsigned rank = 0;
#pragma nounroll
        for (int k = 1; k < 16; k++) {
            const u64 m = __ballot(l == (unsigned)k);
            const unsigned rk = (unsigned)__builtin_amdgcn_readlane((int)runv, k);
            rank = l == (unsigned)k ? rk + (unsigned)__popcll(m & below) : rank;
            runv += lane == k ? (unsigned)__popcll(m) : 0;
        }
        if (l) {
            sorted[rank] = (unsigned short)s;
            if ((int)l <= tb) {
                const unsigned fc = (unsigned)meta[l] + (rank - (unsigned)meta[32 + l]);
                const unsigned rev = __brev(fc) >> (32 - l);
                const unsigned short e = (unsigned short)((s << 4) | l);
                for (unsigned k = rev; k < (1u << tb); k += 1u << l) lut[k] = e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// code longer than the LUT: canonical walk over lengths tb+1..15 (uniform).  -> (symbol << 4 | length) or 0
__device__ __forceinline__ unsigned bz_long_code(unsigned bits, int tb, const unsigned short *sorted, const unsigned short *meta) {
    for (int l = tb + 1; l <= 15; l++) {
        const unsigned code = __brev(bits & ((1u << l) - 1u)) >> (32 - l);
        const unsigned f = bz_rfl(meta[l]), c = bz_rfl(meta[16 + l]);
        if (code - f < c) {
            const unsigned sym = bz_rfl(sorted[bz_rfl(meta[32 + l]) + code - f]);
            return (sym << 4) | (unsigned)l;
        }
    }
    return 0;
}

__global__ __launch_bounds__(64 * BZ_WAVES) void bgzf_inflate(const unsigned char *__restrict__ comp, const BzDesc *__restrict__ blocks,
                                                              int nblocks, unsigned char *__restrict__ out,
                                                              unsigned *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[BZ_WAVES][BZ_LDS];
    const int lane = threadIdx.x & 63;
    const int wv = (int)bz_rfl(threadIdx.x >> 6);
    const int b = blockIdx.x * BZ_WAVES + wv;
    if (b >= nblocks) return;
    unsigned char *lds = lds_all[wv];
    unsigned char *lens = lds;
    unsigned short *lut_ll = (unsigned short *)(lds + BZ_OFF_LUTLL), *lut_d = (unsigned short *)(lds + BZ_OFF_LUTD);
    unsigned short *sorted_ll = (unsigned short *)(lds + BZ_OFF_SORTLL), *sorted_d = (unsigned short *)(lds + BZ_OFF_SORTD);
    unsigned short *meta_ll = (unsigned short *)(lds + BZ_OFF_METALL), *meta_d = (unsigned short *)(lds + BZ_OFF_METAD);

    const BzDesc D = blocks[b];
    const unsigned isize = D.isize, in_len = D.in_len;
    unsigned char *const dst = out + D.out_off;
    // bit input: dword-aligned base, windows of 64 dwords
    const unsigned *const base = (const unsigned *)(comp + (D.in_off & ~3ull));
    const unsigned lead = (unsigned)(D.in_off & 3ull);          // bytes of the first dword that precede the payload
    const unsigned in_dwords = (lead + in_len + 3) / 4;
    unsigned wi = 0;                                           // current window; even windows live in wa, odd ones in wb
    unsigned wa = base[lane], wb = base[64 + lane];
    unsigned di = 0;                                           // next dword to enter the bit buffer
    u64 bb = 0;
    unsigned bc = 0;
    unsigned err = BZ_OK;
    unsigned op = 0;                                           // bytes written
    unsigned nlit = 0, litv = 0;                               // pending literals: lane k holds literal k

#define BZ_FETCH(dst_)                                                                   \
    do {                                                                                 \
        const unsigned win_ = di >> 6;                                                   \
        if (win_ != wi) {                   /* sequential: win_ was prefetched; fetch win_+1 into the register it vacates */ \
            if (di > in_dwords + 2) err = BZ_E_INPUT;   /* sticky; decoding runs dry against ISIZE, nothing more is loaded */ \
            else if (win_ & 1) wa = base[(size_t)(win_ + 1) * 64 + lane];                \
            else wb = base[(size_t)(win_ + 1) * 64 + lane];                              \
            wi = win_;                                                                   \
        }                                                                                \
        {                                                                                \
            const unsigned a_ = (unsigned)__builtin_amdgcn_readlane((int)wa, (int)(di & 63)); \
            const unsigned b_ = (unsigned)__builtin_amdgcn_readlane((int)wb, (int)(di & 63)); \
            dst_ = (win_ & 1) ? b_ : a_;                                                 \
        }                                                                                \
        di++;                                                                            \
    } while (0)
#define BZ_REFILL()                                                                      \
    do {                                                                                 \
        if (bc < 32) {                                                                   \
            unsigned d_;                                                                 \
            BZ_FETCH(d_);                                                                \
            bb |= (u64)d_ << bc;                                                         \
            bc += 32;                                                                    \
        }                                                                                \
    } while (0)
#define BZ_TAKE(n_) (bb >>= (n_), bc -= (n_))
#define BZ_FLUSH_LITS()                                                                  \
    do {                                                                                 \
        if (nlit) {                                                                      \
            if ((unsigned)lane < nlit) dst[op + lane] = (unsigned char)litv;             \
            op += nlit;                                                                  \
            nlit = 0;                                                                    \
        }                                                                                \
    } while (0)

    // drop the bytes in front of the payload
    BZ_REFILL();
    BZ_TAKE(8 * lead);

    bool last = false;
    while (!last && err == BZ_OK) {
        BZ_REFILL();
        last = bb & 1;
        const unsigned btype = (unsigned)(bb >> 1) & 3;
        BZ_TAKE(3);
        if (btype == 0) {  // stored: byte-align, LEN, NLEN, raw bytes
            BZ_TAKE(bc & 7);
            BZ_REFILL();
            const unsigned len = (unsigned)bb & 0xffff, nlen = (unsigned)(bb >> 16) & 0xffff;
            BZ_TAKE(32);
            if ((len ^ nlen) != 0xffff) {
                err = BZ_E_STORED;
                break;
            }
            BZ_FLUSH_LITS();
            const unsigned bytepos = di * 4 - bc / 8;          // from `base`
            if (op + len > isize || bytepos + len > lead + in_len) {
                err = BZ_E_OVERRUN;
                break;
            }
            const unsigned char *src = (const unsigned char *)base + bytepos;
            for (unsigned i = lane; i < len; i += 64) dst[op + i] = src[i];
            op += len;
            const unsigned np = bytepos + len;                  // re-seat the bit reader
            di = np >> 2;
            bb = 0;
            bc = 0;
            wi = di >> 6;                                       // reload both windows at the new position
            {
                const unsigned ea = (wi & 1) ? wi + 1 : wi, eb = (wi & 1) ? wi : wi + 1;
                wa = base[(size_t)ea * 64 + lane];
                wb = base[(size_t)eb * 64 + lane];
            }
            BZ_REFILL();
            BZ_TAKE(8 * (np & 3));
            continue;
        }
        if (btype == 3) {
            err = BZ_E_BTYPE;
            break;
        }
        __builtin_amdgcn_wave_barrier();
        if (btype == 1) {  // fixed code: lengths per RFC 1951 3.2.6
            for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) lens[288 + lane] = lane < 30 ? 5 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!bz_build(lens, 288, BZ_TB_LL, lut_ll, sorted_ll, meta_ll, lane) ||
                !bz_build(lens + 288, 32, BZ_TB_D, lut_d, sorted_d, meta_d, lane)) {
                err = BZ_E_TABLE;
                break;
            }
        } else {  // dynamic code
            BZ_REFILL();
            const unsigned hlit = ((unsigned)bb & 31) + 257, hdist = ((unsigned)(bb >> 5) & 31) + 1, hclen = ((unsigned)(bb >> 10) & 15) + 4;
            BZ_TAKE(14);
            if (hlit > 286 || hdist > 30) {
                err = BZ_E_TABLE;
                break;
            }
            if (lane < 19) lens[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            for (unsigned i = 0; i < hclen; i++) {
                BZ_REFILL();
                if (lane == 0) lens[bz_clorder[i]] = (unsigned char)(bb & 7);
                BZ_TAKE(3);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the code-length code: 7-bit LUT in the distance table's space
            if (!bz_build(lens, 19, 7, lut_d, sorted_d, meta_d, lane)) {
                err = BZ_E_TABLE;
                break;
            }
            unsigned n = 0, prev = 0;
            const unsigned total = hlit + hdist;
            while (n < total && err == BZ_OK) {
                BZ_REFILL();
                const unsigned e = bz_rfl(lut_d[(unsigned)bb & 127]);
                const unsigned l = e & 15, sym = e >> 4;
                if (l == 0) {
                    err = BZ_E_TABLE;
                    break;
                }
                BZ_TAKE(l);
                unsigned rep = 1, val = sym;
                if (sym == 16) {
                    if (n == 0) {
                        err = BZ_E_TABLE;
                        break;
                    }
                    val = prev;
                    rep = 3 + ((unsigned)bb & 3);
                    BZ_TAKE(2);
                } else if (sym == 17) {
                    val = 0;
                    rep = 3 + ((unsigned)bb & 7);
                    BZ_TAKE(3);
                } else if (sym == 18) {
                    val = 0;
                    rep = 11 + ((unsigned)bb & 127);
                    BZ_TAKE(7);
                }
                if (n + rep > total) {
                    err = BZ_E_TABLE;
                    break;
                }
                // lens[0..19) still holds the code-length code: decode into the (not yet built) LUT_LL area, move afterwards
                for (unsigned i = lane; i < rep; i += 64) ((unsigned char *)lut_ll)[n + i] = (unsigned char)val;
                n += rep;
                prev = val;
            }
            if (err != BZ_OK) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {   // move: lens[0..288) = ll lengths (zero padded), lens[288..320) = distance lengths (zero padded)
                const unsigned char *tmp = (const unsigned char *)lut_ll;
                unsigned char v[5], dv = 0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const unsigned s = (unsigned)(k * 64 + lane);
                    v[k] = s < hlit ? tmp[s] : 0;
                }
                if (lane < 32) dv = (unsigned)lane < hdist ? tmp[hlit + lane] : 0;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const unsigned s = (unsigned)(k * 64 + lane);
                    if (s < 288) lens[s] = v[k];
                }
                if (lane < 32) lens[288 + lane] = dv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (bz_rfl(lens[256]) == 0 || !bz_build(lens, 288, BZ_TB_LL, lut_ll, sorted_ll, meta_ll, lane) ||
                !bz_build(lens + 288, 32, BZ_TB_D, lut_d, sorted_d, meta_d, lane)) {
                err = BZ_E_TABLE;
                break;
            }
        }
        // ---- symbols of this block
        for (;;) {
            BZ_REFILL();
            unsigned e = bz_rfl(lut_ll[(unsigned)bb & ((1u << BZ_TB_LL) - 1)]);
            if ((e & 15) == 0) {
                e = bz_long_code((unsigned)bb, BZ_TB_LL, sorted_ll, meta_ll);
                if (e == 0) {
                    err = BZ_E_SYMBOL;
                    break;
                }
            }
            BZ_TAKE(e & 15);
            const unsigned sym = e >> 4;
            if (sym < 256) {
#ifdef BZ_STATS
                if (lane == 0) atomicAdd(&bz_stats[0], 1ull);
#endif
                litv = (unsigned)lane == nlit ? sym : litv;
                nlit++;
                if (nlit == 64) {
                    if (op + 64 > isize) {
                        err = BZ_E_OVERRUN;
                        break;
                    }
                    dst[op + lane] = (unsigned char)litv;
                    op += 64;
                    nlit = 0;
                }
                continue;
            }
            if (sym == 256) break;
            const unsigned lc = sym - 257;
            if (lc > 28) {
                err = BZ_E_SYMBOL;
                break;
            }
            unsigned len;
            if (lc < 8) len = 3 + lc;
            else if (lc == 28) len = 258;
            else {
                const unsigned eb = (lc - 4) >> 2;
                len = 3 + ((4 + (lc & 3)) << eb) + ((unsigned)bb & ((1u << eb) - 1));
                BZ_TAKE(eb);
            }
            BZ_REFILL();
            unsigned ed = bz_rfl(lut_d[(unsigned)bb & ((1u << BZ_TB_D) - 1)]);
            if ((ed & 15) == 0) {
                ed = bz_long_code((unsigned)bb, BZ_TB_D, sorted_d, meta_d);
                if (ed == 0) {
                    err = BZ_E_DIST;
                    break;
                }
            }
            BZ_TAKE(ed & 15);
            const unsigned dc = ed >> 4;
            if (dc > 29) {
                err = BZ_E_DIST;
                break;
            }
            unsigned dist;
            if (dc < 4) dist = 1 + dc;
            else {
                const unsigned eb = (dc >> 1) - 1;
                dist = 1 + ((2 + (dc & 1)) << eb) + ((unsigned)bb & ((1u << eb) - 1));
                BZ_TAKE(eb);
            }
#ifdef BZ_STATS
            if (lane == 0) {
                atomicAdd(&bz_stats[1], 1ull);
                atomicAdd(&bz_stats[2], (unsigned long long)len);
                atomicAdd(&bz_stats[8 + (32 - __clz(dist))], 1ull);
                atomicAdd(&bz_stats[32 + (32 - __clz(len))], 1ull);
            }
#endif
            BZ_FLUSH_LITS();
            if (dist > op) {
                err = BZ_E_DIST;
                break;
            }
            if (op + len > isize) {
                err = BZ_E_OVERRUN;
                break;
            }
            const unsigned char *src = dst + op - dist;
            if (dist >= len) {
                for (unsigned i = lane; i < len; i += 64) dst[op + i] = src[i];
            } else if (dist == 1) {
                const unsigned char v = src[0];
                for (unsigned i = lane; i < len; i += 64) dst[op + i] = v;
            } else {
                for (unsigned i = lane; i < len; i += 64) dst[op + i] = src[i % dist];
            }
            op += len;
        }
    }
    if (err == BZ_OK) {
        BZ_FLUSH_LITS();
        if (op != isize) err = BZ_E_SIZE;
    }
    if (lane == 0) status[b] = err;
#undef BZ_FETCH
#undef BZ_REFILL
#undef BZ_TAKE
#undef BZ_FLUSH_LITS
}

// ---- CRC32 (IEEE, reflected) of every inflated block: lane-parallel segments + GF(2) combine ----------------------
__device__ __forceinline__ unsigned crc_mulmod(unsigned a, unsigned b) {   // a*b mod P, reflected representation (x^0 = bit 31)
    unsigned r = 0;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        r ^= (b & 0x80000000u) ? a : 0;
        a = (a >> 1) ^ ((a & 1) ? 0xEDB88320u : 0);
        b <<= 1;
    }
    return r;
}

struct __attribute__((packed, aligned(1))) BzU128 {
    unsigned w[4];
};

// x^(8 * 1024 * 2^k) mod P for k = 0..5: the tree's multipliers for the segment length of every full BGZF block (65 280 bytes -> 1 024 per lane),
// computed once on the host (tdt_bz_launch) instead of by 17 squarings in every wave
struct CrcPowers {
    unsigned pw[6];
};

__global__ __launch_bounds__(256) void bgzf_crc32(const BzDesc *__restrict__ blocks, int nblocks, const unsigned char *__restrict__ out,
                                                  unsigned *__restrict__ status, CrcPowers P1024) {
    __shared__ unsigned table[4][256];                        // slicing-by-4: table[k][b] = CRC of byte b followed by k zero bytes
    {
        unsigned c = threadIdx.x;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0xEDB88320u : 0);
        table[0][threadIdx.x] = c;
        __syncthreads();
        for (int k = 1; k < 4; k++) {
            c = (c >> 8) ^ table[0][c & 0xff];
            table[k][threadIdx.x] = c;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblocks) return;
    const BzDesc D = blocks[b];
    if (status[b] != BZ_OK) return;
    const unsigned n = D.isize;
    // Lane l owns virtual bytes [l*seg, (l+1)*seg) of (pad zero bytes ++ data): a raw CRC register that starts at 0 is a
    // fixed point of leading zeros, and the register is set to ~0 on reaching data byte 0 (the standard initial value).
    const unsigned seg = ((n + 63) / 64 + 15) & ~15u;          // bytes per lane, a multiple of 16: whole 16-byte loads
    const long long pad = (long long)seg * 64 - n;
    const unsigned char *p = out + D.out_off;
    unsigned c = 0;
    const long long v0 = (long long)lane * seg - pad;
#define CRC_WORD(w_)                                                                                                  \
    do {                                                                                                              \
        c ^= (w_);                                                                                                    \
        c = table[3][c & 0xff] ^ table[2][(c >> 8) & 0xff] ^ table[1][(c >> 16) & 0xff] ^ table[0][c >> 24];          \
    } while (0)
    unsigned i = 0;
    while (i < seg) {
        const long long j = v0 + i;
        if (j + 16 <= 0) {
            i += 16;
            continue;
        }
        if (j >= 0 && i + 128 <= seg) {
            // 128 bytes at a time, the eight loads issued TOGETHER: the lanes of a wave read 1 KB apart, so one load instruction touches
            // 64 cache lines and uses 16 bytes of each; with a load per 16 bytes processed the line had left the L1 (32 waves x 8 KB of
            // such lines per CU) before the lane came back for its next 16 bytes, and every line crossed the L2 eight times
            if (j == 0) c = 0xffffffffu;
            BzU128 v[8];
#pragma unroll
            for (int g = 0; g < 8; g++) v[g] = *(const BzU128 *)(p + j + 16 * g);
#pragma unroll
            for (int g = 0; g < 8; g++) {
#pragma unroll
                for (int k = 0; k < 4; k++) CRC_WORD(v[g].w[k]);
            }
            i += 128;
            continue;
        }
        if (j >= 0) {
            if (j == 0) c = 0xffffffffu;
            const BzU128 v = *(const BzU128 *)(p + j);
#pragma unroll
            for (int k = 0; k < 4; k++) CRC_WORD(v.w[k]);
        } else {                                               // the group that contains data byte 0
            for (int k = 0; k < 16; k++) {
                const long long jj = j + k;
                if (jj < 0) continue;
                if (jj == 0) c = 0xffffffffu;
                c = table[0][(c ^ p[jj]) & 0xff] ^ (c >> 8);
            }
        }
        i += 16;
    }
#undef CRC_WORD
    // the tree: crc(A||B) = crc(A) * x^(8|B|) + crc(B); its multipliers x^(8*seg*2^k) come from the host for seg = 1024 and from
    // square-and-multiply otherwise (a file's last block, short blocks)
    unsigned pw = 0x80000000u;
    if (seg != 1024u) {
        unsigned sq = 0x00800000u;                            // 1 and x^8
        for (unsigned e = seg; e; e >>= 1) {
            if (e & 1) pw = crc_mulmod(pw, sq);
            sq = crc_mulmod(sq, sq);
        }
    }
    int k = 0;
    for (int d = 1; d < 64; d <<= 1, k++) {
        const unsigned right = (unsigned)__shfl_down((int)c, d);
        const unsigned m = seg == 1024u ? P1024.pw[k] : pw;
        if ((lane & (2 * d - 1)) == 0) c = crc_mulmod(c, m) ^ right;
        if (seg != 1024u) pw = crc_mulmod(pw, pw);
    }
    if (lane == 0) {
        const unsigned crc = n ? ~c : 0u;
        if (crc != D.crc) status[b] = BZ_E_CRC;
    }
}

static unsigned crc_mulmod_host(unsigned a, unsigned b) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) {
        r ^= (b & 0x80000000u) ? a : 0;
        a = (a >> 1) ^ ((a & 1) ? 0xEDB88320u : 0);
        b <<= 1;
    }
    return r;
}

static CrcPowers crc_powers_1024() {
    static const CrcPowers P = [] {
        CrcPowers q;
        unsigned pw = 0x80000000u, sq = 0x00800000u;          // 1 and x^8
        for (unsigned e = 1024; e; e >>= 1) {
            if (e & 1) pw = crc_mulmod_host(pw, sq);
            sq = crc_mulmod_host(sq, sq);
        }
        for (int k = 0; k < 6; k++) {
            q.pw[k] = pw;
            pw = crc_mulmod_host(pw, pw);
        }
        return q;
    }();
    return P;
}

__global__ void bgzf_status_reduce(const unsigned *__restrict__ status, int nblocks, unsigned *__restrict__ summary) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblocks && status[i] != BZ_OK) {
        atomicMin(&summary[0], (unsigned)i);                  // first failing block
        atomicAdd(&summary[1], 1u);
    }
}

const char *tdt_bz_err_name(unsigned e) {
    static const char *names[] = {"ok", "reserved block type", "stored-block length check", "invalid Huffman table", "invalid literal/length code",
                                  "invalid distance", "output exceeds ISIZE", "input exhausted", "output shorter than ISIZE", "CRC32 mismatch"};
    return e < 10 ? names[e] : "unknown";
}

// Device-resident form: d_comp holds `comp_len` bytes of whole BGZF blocks followed by >= 1024 readable bytes of padding,
// d_blocks the block table; inflates into d_out and verifies CRC32.  Leaves the per-block status in scratch.
void tdt_bz_launch_lanes(hipStream_t st, int num_cu, int reserve, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks,
                         unsigned char *d_out, unsigned *d_status, unsigned *d_next_block);   // tdt_inflate2.hip

// `st`: the stream the kernels go to; `reserve`: workgroups per CU the persistent inflate grid leaves free (0: the grid is what the chip
// holds — right when nothing else has to run beside it; the ingest, whose record search / decode / consumer kernels run on another stream
// while the next span inflates, keeps one)
int tdt_bz_launch_on(tdt_ctx *ctx, hipStream_t st, int reserve, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks,
                     unsigned char *d_out, bool check_crc, unsigned *d_status, unsigned *d_summary) {
    TDT_HIP(hipMemsetAsync(d_summary, 0xff, 4, st));
    TDT_HIP(hipMemsetAsync(d_summary + 1, 0, 8, st));            // (+ the lanes kernel's block counter, d_summary[2])
    const unsigned grid = (unsigned)((nblocks + BZ_WAVES - 1) / BZ_WAVES);
    const bool sequential = getenv("TIDDIT_INFLATE_SEQ") != nullptr;   // the one-symbol-at-a-time kernel of this file
    if (sequential) hipLaunchKernelGGL(bgzf_inflate, dim3(grid), dim3(64 * BZ_WAVES), 0, st, d_comp, d_blocks, (int)nblocks, d_out, d_status);
    else tdt_bz_launch_lanes(st, ctx->num_cu, reserve, d_comp, d_blocks, nblocks, d_out, d_status, d_summary + 2);
    TDT_CHECK_LAUNCH();
    if (check_crc) {
        hipLaunchKernelGGL(bgzf_crc32, dim3((unsigned)((nblocks + 3) / 4)), dim3(256), 0, st, d_blocks, (int)nblocks, d_out, d_status, crc_powers_1024());
        TDT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bgzf_status_reduce, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, d_status, (int)nblocks, d_summary);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

int tdt_bz_launch(tdt_ctx *ctx, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks, unsigned char *d_out, bool check_crc,
                     unsigned *d_status, unsigned *d_summary) {
    return tdt_bz_launch_on(ctx, ctx->stream, 0, d_comp, d_blocks, nblocks, d_out, check_crc, d_status, d_summary);
}


int tdt_bz_block_table(const uint8_t *comp, size_t len, std::vector<BzDesc> &blocks, size_t *produced) {
    size_t o = 0, uo = 0;
    while (o < len) {
        size_t bs, po, pl;
        uint32_t isz;
        if (tdt_bz_hop(comp + o, len - o, &bs, &po, &pl, &isz) != 1 || isz > 65536) {
            tdt_set_error("tdt_bgzf_inflate_hbm: input is not a whole number of BGZF blocks (offset %zu)", o);
            return TDT_E_ARG;
        }
        const uint8_t *c = comp + o + bs - 8;
        blocks.push_back(BzDesc{o + po, uo, (unsigned)pl, isz, (unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16) | ((unsigned)c[3] << 24), 0});
        o += bs;
        uo += isz;
    }
    *produced = uo;
    return TDT_OK;
}

extern "C" int tdt_bgzf_inflate_hbm(tdt_ctx *ctx, const uint8_t *comp, size_t len, uint8_t *out, size_t out_len, int out_on_device) {
    if (!ctx || (!comp && len) || (!out && out_len)) {
        tdt_set_error("tdt_bgzf_inflate_hbm: bad argument");
        return TDT_E_ARG;
    }
    std::vector<BzDesc> blocks;
    size_t produced = 0;
    int rc = tdt_bz_block_table(comp, len, blocks, &produced);
    if (rc) return rc;
    if (produced != out_len) {
        tdt_set_error("tdt_bgzf_inflate_hbm: blocks inflate to %zu bytes, caller gave %zu", produced, out_len);
        return TDT_E_ARG;
    }
    if (blocks.empty()) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nb = blocks.size();
    const size_t comp_pad = ((len + 4096 + 255) & ~(size_t)255), tab = ((nb * sizeof(BzDesc) + 255) & ~(size_t)255);
    void *d = nullptr;
    rc = tdt_scratch(ctx, 15, comp_pad + tab + ((nb * 4 + 255) & ~(size_t)255) + 256, &d);
    if (rc) return rc;
    unsigned char *d_comp = (unsigned char *)d;
    BzDesc *d_blocks = (BzDesc *)(d_comp + comp_pad);
    unsigned *d_status = (unsigned *)((char *)d_blocks + tab);
    unsigned *d_summary = (unsigned *)((char *)d_status + ((nb * 4 + 255) & ~(size_t)255));
    unsigned char *d_out = out;
    if (!out_on_device) {
        void *o = nullptr;
        rc = tdt_scratch(ctx, 16, out_len + 256, &o);
        if (rc) return rc;
        d_out = (unsigned char *)o;
    }
    TDT_HIP(hipMemcpyAsync(d_comp, comp, len, hipMemcpyHostToDevice, st));
    TDT_HIP(hipMemsetAsync(d_comp + len, 0, comp_pad - len, st));
    TDT_HIP(hipMemcpyAsync(d_blocks, blocks.data(), nb * sizeof(BzDesc), hipMemcpyHostToDevice, st));
    rc = tdt_bz_launch(ctx, d_comp, d_blocks, nb, d_out, true, d_status, d_summary);
    if (rc) return rc;
    unsigned summary[2] = {0, 0};
    TDT_HIP(hipMemcpyAsync(summary, d_summary, 8, hipMemcpyDeviceToHost, st));
    if (!out_on_device) TDT_HIP(hipMemcpyAsync(out, d_out, out_len, hipMemcpyDeviceToHost, st));
    TDT_HIP(hipStreamSynchronize(st));
    if (summary[1]) {
        unsigned code = 0;
        TDT_HIP(hipMemcpy(&code, d_status + summary[0], 4, hipMemcpyDeviceToHost));
        tdt_set_error("tdt_bgzf_inflate_hbm: %u of %zu blocks failed; first is block %u: %s", summary[1], nb, summary[0], tdt_bz_err_name(code));
        return TDT_E_ARG;
    }
    return TDT_OK;
}
