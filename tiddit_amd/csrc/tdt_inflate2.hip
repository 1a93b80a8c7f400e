// BGZF inflate, lane-parallel symbol decode (the default kernel; tdt_inflate.hip keeps the one-symbol-at-a-time version).
//
// Still one wavefront per BGZF block, but inside a Huffman-coded DEFLATE block the 64 lanes decode SPECULATIVELY: lane i
// decodes the symbol that would start i bits into the current window — literal/length LUT, and for a length code its
// extra bits, the distance LUT and the distance extra bits — which yields next[i] = i + bits consumed.  The true symbol
// sequence is the chain 0 -> next[0] -> next[next[0]] ... ; a short scalar walk (one v_readlane per symbol) marks the
// lanes on it, a DPP prefix sum of their output lengths gives every symbol its output offset, all literals of the window
// leave in ONE predicated store and only the matches (LZ77 copies, which depend on earlier output) are replayed in
// order.  A 64-bit window holds ~7 symbols, so the per-symbol scalar work — what bounds the sequential kernel — drops
// about threefold, and the table lookups/bit arithmetic move to the otherwise idle vector pipe.
//   * input: the compressed stream is staged through a 512-byte LDS ring (two 256-byte windows, one coalesced load each);
//     lanes gather their three dwords from it;
//   * tables: canonical Huffman LUTs in LDS, 32-bit entries with RFC 1951's base value / extra-bit count folded in
//     (9 bits literal/length, 8 bits distance); a symbol whose code is longer than the LUT stops the chain and is decoded
//     by the scalar canonical walk;
//   * every loop is bounded by ISIZE / the compressed length; damage sets the block's status.
// ---- measurement builds declare themselves (tdt_build_flags): the macros this file was compiled with, before any default is set
extern const char *const tdt_variant_inflate2;
const char *const tdt_variant_inflate2 = ""
#ifdef B2_EARLYM
    " B2_EARLYM"
#endif
#ifdef B2_EXP_NOLIT
    " B2_EXP_NOLIT"
#endif
#ifdef B2_EXP_NOOWN
    " B2_EXP_NOOWN"
#endif
#ifdef B2_EXP_NOREPLAY
    " B2_EXP_NOREPLAY"
#endif
#ifdef B2_HOP2
    " B2_HOP2"
#endif
#ifdef B2_NOPERSIST
    " B2_NOPERSIST"
#endif
#ifdef B2_OCC
    " B2_OCC"
#endif
#ifdef B2_PHASED
    " B2_PHASED"
#endif
#ifdef B2_PIPE
    " B2_PIPE"
#endif
#ifdef B2_PROF
    " B2_PROF"
#endif
#ifdef B2_STATS
    " B2_STATS"
#endif
#ifdef B2_TB_D
    " B2_TB_D"
#endif
#ifdef B2_TB_LL
    " B2_TB_LL"
#endif
#ifdef B2_WAVES
    " B2_WAVES"
#endif
#ifdef B2_WBITS
    " B2_WBITS"
#endif
#ifdef B2_W2
    " B2_W2"
#endif
#ifdef B2_W2_EARLY
    " B2_W2_EARLY"
#endif
    ;

#include "tdt_common.h"

#include <algorithm>

// LUT widths and the occupancy target: 9/8 bits keep a wave's LDS at 4.7 KB and, with registers held to 64, eight waves per
// SIMD — measured 40 ms per 1.94 GB against 51 ms for 10/9 bits at five waves (latency hiding beats the rarer long-code path)
#ifndef B2_TB_LL
#define B2_TB_LL 9
#endif
#ifndef B2_TB_D
#define B2_TB_D 8
#endif
#ifndef B2_OCC
#define B2_OCC 8
#endif
#ifndef B2_WAVES
#define B2_WAVES 4
#endif
#define B2_PARMAX 16                       // longest match a lane copies by itself (bytes); 32 with two loads was measured: slower
#ifndef B2_HOP2
#define B2_HOP2 1
#endif
#ifndef B2_W2
#define B2_W2 0                            // 1: TWO windows (128 bit offsets) per trip of the symbol loop — see the B2_W2 loop below
#endif
#ifndef B2_W2_EARLY
#define B2_W2_EARLY 0                      // B2_W2: 1 = each set's first long match that reads only earlier trips' output is copied by all lanes in the own-lane
                                           // phases (measured: +4.5 % / +7 % SLOWER than B2_W2 without it — the instructions it adds to every trip cost more
                                           // than the replay round trips it saves, as B2_EARLYM in the one-window loop)
#endif
#ifndef B2_WBITS
#define B2_WBITS 64                        // bit offsets a window's chain walk accepts (measurement builds: 32 / 16 — what a window costs apart from its symbols)
#endif
#ifndef B2_PHASED
#define B2_PHASED 0                        // 1: all loads of the window's independent copies first, one wait, then all their stores (measured in round 4:
#endif                                     // 15.0 ms against 14.75 at level 6, 16.9 against 17.4 at level 1 — kept as a variant; B2_EARLYM adds the window's
#ifndef B2_EARLYM                          // first long independent match to the phase: 15.05 / 17.25)
#define B2_EARLYM 0
#endif
#ifndef B2_PIPE
#define B2_PIPE 0                          // 1: the copies pipelined over two windows (measured in round 4: 18.3 ms against 17.9, 22.4 against 21.7 —
#endif                                     // the wait it moves is not what a window waits for; ten more live registers spill).  Kept as a variant.
// LDS bytes per wave: lens 320 | lut_ll 4 << TB_LL | lut_d 4 << TB_D | sorted_ll 576 | sorted_d 64 | meta_ll 96 | meta_d 96 | ring 512 + 16
// (the ring's first two dwords are mirrored behind it: a lane's three consecutive dwords never wrap, one address serves all three reads)
#define B2_OFF_LUTLL 320
#define B2_OFF_LUTD (B2_OFF_LUTLL + (4 << B2_TB_LL))
#define B2_OFF_SORTLL (B2_OFF_LUTD + (4 << B2_TB_D))
#define B2_OFF_SORTD (B2_OFF_SORTLL + 576)
#define B2_OFF_METALL (B2_OFF_SORTD + 64)
#define B2_OFF_METAD (B2_OFF_METALL + 96)
#define B2_OFF_WIN (B2_OFF_METAD + 96)
#define B2_LDS (B2_OFF_WIN + 512 + 16)
// LUT entry (32 bits): 0-3 code length l, 4-7 number of extra bits eb, 8-22 base value (literal byte / base length <= 258 / base
// distance <= 24577 / raw symbol), 20 = length code, 21 = end of block (literal/length table only: its bases need 9 bits), and
// 23-31 STEP: how far the bit cursor moves past this code and its extra bits — l for a literal, l + eb for a length or a distance,
// 256 + l for the end of block, 128 for "longer than the LUT / unused" (B2_ESC).  A lane's next[] is then lane + step (+ the distance
// entry's step behind a length code): no compare, no select; values in [128, 256) stop the chain at a code the tables cannot
// resolve, values >= 256 at the end of the block.
#define B2_ESC 0x40000000u
#define B2_F_LEN (1u << 20)
#define B2_F_EOB (1u << 21)
#define B2_STEP(e) ((e) >> 23)
enum { B2_MODE_RAW = 0, B2_MODE_LL = 1, B2_MODE_DIST = 2 };
enum { B2_OK = 0, B2_E_BTYPE = 1, B2_E_STORED = 2, B2_E_TABLE = 3, B2_E_SYMBOL = 4, B2_E_DIST = 5, B2_E_OVERRUN = 6, B2_E_INPUT = 7, B2_E_SIZE = 8 };

typedef unsigned long long u64;
struct __attribute__((packed, aligned(1))) B2U32 {
    unsigned v;
};
struct __attribute__((packed, aligned(1))) B2U16 {
    unsigned short v;
};
struct __attribute__((packed, aligned(1))) B2U128 {
    unsigned w[4];
};

__device__ __forceinline__ unsigned b2_rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned b2_rl(unsigned v, unsigned lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, (int)lane); }

#ifdef B2_STATS   // measurement builds only (tools/inflate_stats2.py): what the windows of the lanes kernel are made of
__device__ unsigned long long b2_stats[16];   // 0 windows, 1 symbols, 2 literals, 3 matches copied by their own lane, 4 replayed: longer than 16, 5 replayed: source
                                              // inside the window's own output, 6 stops at a long code, 7 match bytes, 8 replayed match bytes
extern "C" int tdt_debug_b2_stats(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(b2_stats), z, sizeof z) == hipSuccess ? 0 : -2;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2_stats), 16 * 8) == hipSuccess ? 0 : -2;
}
#endif

#ifdef B2_PROF   // measurement builds only (tools/inflate_prof.py): shader-clock cycles per phase of the window loop, summed over all waves
__device__ unsigned long long b2_prof[16];    // 0 gather + LUT lookups, 1 per-lane lengths / distances, 2 chain walk (+ long codes), 3 prefix sum + checks,
                                              // 4 literal store + own-lane copies (load, wait, stores), 5 replayed matches, 6 cursor + ring refill, 7 tables / headers, 8 windows,
                                              // 9 the hops of the chain walk alone (2 then holds the long-code path), 10 long codes resolved, 11 DEFLATE block header + tables (7 then: between windows)
extern "C" int tdt_debug_b2_prof(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long z[16] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(b2_prof), z, sizeof z) == hipSuccess ? 0 : -2;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(b2_prof), 16 * 8) == hipSuccess ? 0 : -2;
}
#define B2_MARK(k_)                                               \
    do {                                                          \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        pf[k_] += (unsigned)(t_ - pf_last);                       \
        pf_last = t_;                                             \
    } while (0)
#else
#define B2_MARK(k_) do { } while (0)
#endif

// lane's bit of a 64-bit scalar mask selects between two values: ONE v_cndmask with the mask as its SGPR-pair operand
__device__ __forceinline__ unsigned b2_sel(u64 m, unsigned if_set, unsigned if_clear) {
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}

// one lane of a vector register <- a scalar value (v_writelane_b32; value and lane index are wave-uniform)
__device__ __forceinline__ void b2_wl(unsigned &reg, unsigned value, unsigned lane_index) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(value), "s"(lane_index) : "m0");   // (one SGPR per VOP3 on gfx9: the lane goes through M0)
}

__constant__ unsigned char b2_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ unsigned b2_entry(int mode, unsigned sym, unsigned l) {
    if (mode == B2_MODE_RAW) return (sym << 8) | l;
    if (mode == B2_MODE_LL) {
        if (sym < 256) return (l << 23) | (sym << 8) | l;
        if (sym == 256) return ((256u + l) << 23) | B2_F_EOB | l;
        const unsigned lc = sym - 257;
        if (lc > 28) return B2_ESC;
        unsigned eb = 0, base = 3 + lc;
        if (lc == 28) base = 258;
        else if (lc >= 8) {
            eb = (lc - 4) >> 2;
            base = 3 + ((4 + (lc & 3)) << eb);
        }
        return ((l + eb) << 23) | B2_F_LEN | (base << 8) | (eb << 4) | l;
    }
    if (sym > 29) return B2_ESC;
    unsigned eb = 0, base = 1 + sym;
    if (sym >= 4) {
        eb = (sym >> 1) - 1;
        base = 1 + ((2 + (sym & 1)) << eb);
    }
    return ((l + eb) << 23) | (base << 8) | (eb << 4) | l;
}

// Canonical Huffman tables from `n` code lengths in LDS (lane k carries the state of code length k).
__device__ __forceinline__ bool b2_build(const unsigned char *lens, int n, int tb, int mode, unsigned *lut, unsigned short *sorted,
                                         unsigned short *meta, int lane) {
    for (int i = lane; i < (1 << tb); i += 64) lut[i] = B2_ESC;
    unsigned cntv = 0;
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
        const unsigned ck = b2_rl(cntv, k);
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
    unsigned runv = offv;
    const u64 below = (1ull << lane) - 1ull;
#pragma nounroll
    for (int c = 0; c < n; c += 64) {
        const int s = c + lane;
        const unsigned l = s < n ? lens[s] : 0;
        unsigned rank = 0;
#pragma nounroll
        for (int k = 1; k < 16; k++) {
            const u64 m = __ballot(l == (unsigned)k);
            const unsigned rk = b2_rl(runv, k);
            rank = l == (unsigned)k ? rk + (unsigned)__popcll(m & below) : rank;
            runv += lane == k ? (unsigned)__popcll(m) : 0;
        }
        if (l) {
            sorted[rank] = (unsigned short)s;
            if ((int)l <= tb) {
                const unsigned fc = (unsigned)meta[l] + (rank - (unsigned)meta[32 + l]);
                const unsigned rev = __brev(fc) >> (32 - l);
                const unsigned e = b2_entry(mode, (unsigned)s, l);
                for (unsigned k = rev; k < (1u << tb); k += 1u << l) lut[k] = e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Code longer than the LUT.  Lane l (tb < l <= 15) tests code length l — first code, count and offset of that length come from the
// table's meta block in one round of LDS reads — a ballot picks the shortest length that matches (a prefix code has exactly one), and
// one more read fetches the symbol: two LDS round trips, where the length-by-length walk of rounds 1-3 needed up to three per length
// (one symbol in ten takes this path on BAM-shaped data, and it sits inside the scalar chain walk).  -> LUT entry, B2_ESC if no code matches
__device__ __forceinline__ unsigned b2_long_code(unsigned bits, int tb, int mode, const unsigned short *sorted, const unsigned short *meta, int lane) {
    const unsigned l = (unsigned)lane & 15u;
    const unsigned code = l ? __brev(bits & ((1u << l) - 1u)) >> (32 - l) : 0u;
    const unsigned f = meta[l], c = meta[16 + l], off = meta[32 + l];
    const bool hit = lane < 16 && (int)l > tb && code - f < c;
    const u64 m = __ballot(hit);
    if (!m) return B2_ESC;
    const unsigned L = (unsigned)__builtin_ctzll(m);
    const unsigned idx = b2_rl(off + code - f, L);
    const unsigned sym = b2_rfl(sorted[idx]);
    return b2_entry(mode, sym, L);
}

// inclusive prefix sum over the 64 lanes: six fused DPP adds (Hillis-Steele inside each row of 16, then the two row
// broadcasts); the s_nop pairs cover the VALU-write -> DPP-read hazard the assembler does not see inside inline asm
__device__ __forceinline__ unsigned b2_scan(unsigned v) {
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return v;
}

// two independent inclusive prefix sums over the 64 lanes, their DPP steps interleaved: a step's result is read two instructions
// later (the other value's step + one wait state), so the pair costs little more than one scan
__device__ __forceinline__ void b2_scan2(unsigned &a, unsigned &b) {
#define B2_S2(CTRL) "v_add_u32_dpp %0, %0, %0 " CTRL "\n\tv_add_u32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
    asm volatile("s_nop 1\n\t"
                 B2_S2("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 B2_S2("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 B2_S2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 B2_S2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 B2_S2("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 B2_S2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
#undef B2_S2
}

// B2_W2: what the symbol starting `q` bits into the stream would be, for the lane's own q.  -> pk = the chain walk's packed word (bits 7:0 the
// lane one symbol ahead, 64 when that hop leaves the window; from bit 8 where the walk stands after two symbols), sy = what the symbol
// writes in ONE word: bits 8:0 the number of output bytes (0 end of block, 1 literal, 3..258 match), from bit 9 the literal byte or the
// match's distance - 1
__device__ __forceinline__ void b2_spec(const unsigned *win, const unsigned *lut_ll, const unsigned *lut_d, unsigned q, int lane, unsigned &pk, unsigned &sy) {
    const unsigned qd = (q >> 5) & 127, qs = q & 31;
    const unsigned wa = win[qd], wb = win[qd + 1], wc = win[qd + 2];
    const unsigned lo = __builtin_amdgcn_alignbit(wb, wa, qs), hi = __builtin_amdgcn_alignbit(wc, wb, qs);
    const unsigned e1 = lut_ll[lo & ((1u << B2_TB_LL) - 1)];
    const unsigned step1 = B2_STEP(e1);
    const unsigned e2 = lut_d[__builtin_amdgcn_ubfe(lo, step1, B2_TB_D)];
    const bool is_len = e1 & B2_F_LEN;
    const unsigned nxt = (unsigned)lane + step1 + (is_len ? B2_STEP(e2) : 0u);
    const unsigned m1 = nxt < 64u ? nxt : 64u;
    const unsigned n2r = (unsigned)__builtin_amdgcn_ds_bpermute((int)(m1 << 2), (int)nxt);
    pk = m1 | ((nxt < 64u ? n2r : nxt) << 8);
    const unsigned l1 = e1 & 15, eb1 = (e1 >> 4) & 15, litv = (e1 >> 8) & 0x1ff;
    const unsigned mlen = litv + __builtin_amdgcn_ubfe(lo >> l1, 0u, eb1);
    const unsigned l2 = e2 & 15, eb2 = (e2 >> 4) & 15, base2 = (e2 >> 8) & 0x7fff;
    const unsigned dist = base2 + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi, lo, (step1 + l2) & 31), 0u, eb2);
    sy = is_len ? (mlen | ((dist - 1u) << 9)) : ((e1 & B2_F_EOB) ? 0u : (1u | (litv << 9)));
}

// B2_W2: the chain of true symbol starts through one set of 64 bit offsets, from lane `cur` (scalar; two symbols per hop).  Codes longer
// than the LUTs are resolved on the way and their lane's `sy` patched (as in the one-window loop).  -> chain = the lanes on it, cur = where
// the walk left the set (64 .. 111: that many bits behind the set's first; the bits past an end of block when stop == 2)
__device__ __forceinline__ void b2_walk(const unsigned *win, const unsigned *lut_ll, const unsigned *lut_d, const unsigned short *sorted_ll,
                                        const unsigned short *meta_ll, const unsigned short *sorted_d, const unsigned short *meta_d, unsigned qbase,
                                        int lane, unsigned pk, unsigned &sy, u64 &chain, unsigned &cur, unsigned &stop, unsigned &err) {
    const unsigned start = cur;
    for (;;) {
        while (cur < 64u) {
            const unsigned p2 = b2_rl(pk, cur);
            asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(cur));
            asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(p2));       // (bits 5:0 = the next symbol's lane, or lane 0 when that hop leaves the set: undone below)
            cur = p2 >> 8;
        }
        if (__builtin_expect(cur < 128u, 1)) break;                      // (the common exit first, as in the one-window loop)
        if (cur >= 256u) {
            stop = 2;
            cur -= 256u;
            break;
        }
        const unsigned at = 63u - (unsigned)__builtin_clzll(chain);      // the chain's last member is the symbol the LUTs did not resolve
        const unsigned qq = qbase + at, d = (qq >> 5) & 127, sh = qq & 31;
        const unsigned w0 = b2_rfl(win[d]), w1 = b2_rfl(win[d + 1]), w2 = b2_rfl(win[d + 2]);
        u64 v = ((u64)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
        unsigned e = b2_rfl(lut_ll[(unsigned)v & ((1u << B2_TB_LL) - 1)]);
        if (e == B2_ESC) e = b2_long_code((unsigned)v, B2_TB_LL, B2_MODE_LL, sorted_ll, meta_ll, lane);
        if (e == B2_ESC) {
            err = B2_E_SYMBOL;
            break;
        }
        unsigned used = e & 15;
        v >>= used;
        unsigned s_sy = 1u | (((e >> 8) & 0xff) << 9);
        if (e & B2_F_EOB) {
            s_sy = 0;
            stop = 2;
        } else if (e & B2_F_LEN) {
            unsigned eb = (e >> 4) & 15;
            const unsigned s_len = ((e >> 8) & 0x1ff) + ((unsigned)v & ((1u << eb) - 1));
            v >>= eb;
            used += eb;
            unsigned ed = b2_rfl(lut_d[(unsigned)v & ((1u << B2_TB_D) - 1)]);
            if (ed == B2_ESC) ed = b2_long_code((unsigned)v, B2_TB_D, B2_MODE_DIST, sorted_d, meta_d, lane);
            if (ed == B2_ESC) {
                err = B2_E_DIST;
                break;
            }
            v >>= ed & 15;
            eb = (ed >> 4) & 15;
            const unsigned s_dist = ((ed >> 8) & 0x7fff) + ((unsigned)v & ((1u << eb) - 1));
            used += (ed & 15) + eb;
            s_sy = s_len | ((s_dist - 1u) << 9);
        }
        b2_wl(sy, s_sy, at);
        cur = at + used;
        if (stop == 2) break;
    }
    if (start != 0) chain &= ~1ull;                                 // (a walk that starts behind lane 0 never has lane 0 on its chain)
}

// 64 stream bits starting `q` bits into the stream, gathered from the LDS ring (any lane, any q inside the staged windows)
__device__ __forceinline__ u64 b2_bits_at(const unsigned *win, unsigned q) {
    const unsigned d = (q >> 5) & 127, s = q & 31;
    const unsigned a = win[d], b = win[d + 1], c = win[d + 2];                 // (the mirror behind the ring: no wrap)
    const unsigned lo = __builtin_amdgcn_alignbit(b, a, s), hi = __builtin_amdgcn_alignbit(c, b, s);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64 * B2_WAVES) __attribute__((amdgpu_waves_per_eu(B2_OCC, 8))) void bgzf_inflate_lanes(const unsigned char *__restrict__ comp, const BzDesc *__restrict__ blocks,
                                                                    int nblocks, unsigned char *__restrict__ out,
                                                                    unsigned *__restrict__ status, unsigned *__restrict__ next_block) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[B2_WAVES][B2_LDS];
    const int lane = threadIdx.x & 63;
    const int wv = (int)b2_rfl(threadIdx.x >> 6);
    // Persistent waves: the grid is what the chip holds at once (8 waves per SIMD) and every wave takes BGZF blocks off one counter
    // until none is left.  With one launch slot per block, a span of 20 k blocks ran as 2.5 "rounds" of 8192 resident waves, the last
    // one 60 % empty, and a workgroup's four slots stayed taken until its slowest block was done.
#ifdef B2_NOPERSIST   // measurement variant: one launch slot per block, as before round 4
    for (int once = 0; once < 1; once++) {
    const int b = blockIdx.x * B2_WAVES + wv;
    if (b >= nblocks) return;
#else
    for (;;) {
    unsigned b_ = 0;
    if (lane == 0) b_ = atomicAdd(next_block, 1u);
    const int b = (int)b2_rfl(b_);
    if (b >= nblocks) return;
#endif
    unsigned char *lds = lds_all[wv];
    unsigned char *lens = lds;
    unsigned *lut_ll = (unsigned *)(lds + B2_OFF_LUTLL), *lut_d = (unsigned *)(lds + B2_OFF_LUTD);
    unsigned short *sorted_ll = (unsigned short *)(lds + B2_OFF_SORTLL), *sorted_d = (unsigned short *)(lds + B2_OFF_SORTD);
    unsigned short *meta_ll = (unsigned short *)(lds + B2_OFF_METALL), *meta_d = (unsigned short *)(lds + B2_OFF_METAD);
    unsigned *win = (unsigned *)(lds + B2_OFF_WIN);

    const BzDesc D = blocks[b];
    const unsigned isize = D.isize, in_len = D.in_len;
    unsigned char *const dst = out + D.out_off;
    const unsigned *const base = (const unsigned *)(comp + (D.in_off & ~3ull));
    const unsigned lead = (unsigned)(D.in_off & 3ull);
    const unsigned end_bit = (lead + in_len) * 8;               // first bit past the payload
    unsigned bp = lead * 8;                                     // stream position in bits from `base`
    unsigned cw = 0;                                            // windows cw and cw+1 are staged (window w in ring half w & 1)
    unsigned err = B2_OK;
    unsigned op = 0;
#ifdef B2_PROF
    unsigned pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pf_last = __builtin_readcyclecounter();
#endif

// ring half h (0 / 1) <- the 64 dwords of window w_; the ring's first two dwords are mirrored behind its end
#define B2_STAGE(h_, w_)                                                                   \
    do {                                                                                   \
        const unsigned v_ = base[(size_t)(w_) * 64 + lane];                                \
        win[(h_) * 64 + lane] = v_;                                                        \
        if ((h_) == 0 && lane < 2) win[128 + lane] = v_;                                   \
    } while (0)
    B2_STAGE(0u, 0u);
    B2_STAGE(1u, 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

// keep the windows that hold bp .. bp + 160 bits staged; a position beyond the payload ends the decode (sticky error)
#define B2_ENSURE()                                                                       \
    do {                                                                                  \
        const unsigned w_ = bp >> 11;                                                     \
        if (w_ != cw) {                                                                   \
            if (bp > end_bit + 64) err = B2_E_INPUT;                                      \
            else {                                                                        \
                __builtin_amdgcn_wave_barrier();                                          \
                if (w_ != cw + 1) B2_STAGE(w_ & 1u, w_);                                  \
                B2_STAGE((w_ + 1u) & 1u, w_ + 1u);                                        \
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                    \
                __builtin_amdgcn_wave_barrier();                                          \
            }                                                                             \
            cw = w_;                                                                      \
        }                                                                                 \
    } while (0)

#if B2_PIPE
    // copies of the previous window that are still to be stored: per lane its kind (1 = own-lane copy with the source bytes loaded,
    // 2 = own-lane byte copy, 3 = replayed in stream order by all lanes), where, how long, how far back, and the loaded bytes
    unsigned pd_kind = 0, pd_pos = 0, pd_mlen = 0, pd_dist = 0, pd_tail = 0;
    B2U128 pd_v = {{0, 0, 0, 0}};
#define B2_FLUSH()                                                                                               \
    do {                                                                                                         \
        if (pd_kind == 2u)                      /* a 3-byte match, or a source in the block's last bytes */      \
            for (unsigned k_ = 0; k_ < pd_mlen; k_++) dst[pd_pos + k_] = dst[pd_pos - pd_dist + k_];             \
        if (pd_kind == 1u) {                    /* whole dwords, the last one overlapping its predecessor */     \
            unsigned char *const p_ = dst + pd_pos;                                                              \
            reinterpret_cast<B2U32 *>(p_)->v = pd_v.w[0];                                                        \
            if (pd_mlen >= 8) reinterpret_cast<B2U32 *>(p_ + 4)->v = pd_v.w[1];                                  \
            if (pd_mlen >= 12) reinterpret_cast<B2U32 *>(p_ + 8)->v = pd_v.w[2];                                 \
            if (pd_mlen >= 16) reinterpret_cast<B2U32 *>(p_ + 12)->v = pd_v.w[3];                                \
            if (pd_mlen & 3) reinterpret_cast<B2U32 *>(p_ + pd_mlen - 4)->v = pd_tail;                           \
        }                                                                                                        \
        u64 mm_ = __ballot(pd_kind == 3u);      /* the others in stream order (they may read each other's output) */ \
        while (mm_) {                                                                                            \
            const unsigned l_ = (unsigned)__builtin_ctzll(mm_);                                                  \
            mm_ &= ~(1ull << l_);                                                                                \
            const unsigned len_ = b2_rl(pd_mlen, l_), dd_ = b2_rl(pd_dist, l_), q_ = b2_rl(pd_pos, l_), so_ = q_ - dd_; \
            unsigned i_ = (unsigned)lane;                                                                        \
            do {                                                                                                 \
                const unsigned j_ = dd_ >= len_ ? i_ : i_ % dd_;   /* a distance shorter than the match repeats its source */ \
                if (i_ < len_) dst[q_ + i_] = dst[so_ + j_];                                                     \
                i_ += 64;                                                                                        \
            } while (i_ - (unsigned)lane < len_);                                                                \
        }                                                                                                        \
        pd_kind = 0;                                                                                             \
    } while (0)
#endif
    bool last = false;
    while (!last && err == B2_OK) {
        u64 hb;
        {
            const unsigned d = bp >> 5, s = bp & 31;
            hb = (((u64)b2_rfl(win[(d + 1) & 127]) << 32) | b2_rfl(win[d & 127])) >> s;
        }
        last = hb & 1;
        const unsigned btype = (unsigned)(hb >> 1) & 3;
        bp += 3;
        if (btype == 0) {  // stored: byte-align, LEN, NLEN, raw bytes straight from the compressed buffer
            bp = (bp + 7) & ~7u;
            B2_ENSURE();
            const unsigned d = bp >> 5, s = bp & 31;
            const u64 v = (((u64)b2_rfl(win[(d + 1) & 127]) << 32) | b2_rfl(win[d & 127])) >> s;
            const unsigned len = (unsigned)v & 0xffff, nlen = (unsigned)(v >> 16) & 0xffff;
            bp += 32;
            if ((len ^ nlen) != 0xffff) {
                err = B2_E_STORED;
                break;
            }
            if (op + len > isize || bp / 8 + len > lead + in_len) {
                err = B2_E_OVERRUN;
                break;
            }
            const unsigned char *src = (const unsigned char *)base + bp / 8;
            for (unsigned i = lane; i < len; i += 64) dst[op + i] = src[i];
            op += len;
            bp += len * 8;
            B2_ENSURE();
            continue;
        }
        if (btype == 3) {
            err = B2_E_BTYPE;
            break;
        }
        __builtin_amdgcn_wave_barrier();
        if (btype == 1) {  // fixed code: lengths per RFC 1951 3.2.6
            for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) lens[288 + lane] = lane < 30 ? 5 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {  // dynamic code: HLIT, HDIST, HCLEN, the code-length code, then the run-length coded lengths
            B2_ENSURE();
            unsigned hlit, hdist, hclen;
            {
                const unsigned d = bp >> 5, s = bp & 31;
                const u64 v = (((u64)b2_rfl(win[(d + 1) & 127]) << 32) | b2_rfl(win[d & 127])) >> s;
                hlit = ((unsigned)v & 31) + 257;
                hdist = ((unsigned)(v >> 5) & 31) + 1;
                hclen = ((unsigned)(v >> 10) & 15) + 4;
                bp += 14;
            }
            if (hlit > 286 || hdist > 30) {
                err = B2_E_TABLE;
                break;
            }
            {   // 3 bits per code-length-code length: lane i reads its own field
                const unsigned q = bp + 3 * (unsigned)lane;
                const unsigned v = (unsigned)b2_bits_at(win, (unsigned)lane < hclen ? q : bp) & 7;
                if (lane < 19) lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                if ((unsigned)lane < hclen) lens[b2_clorder[lane]] = (unsigned char)v;
                bp += 3 * hclen;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            B2_ENSURE();
            if (!b2_build(lens, 19, 7, B2_MODE_RAW, lut_d, sorted_d, meta_d, lane)) {
                err = B2_E_TABLE;
                break;
            }
            unsigned n = 0, prev = 0;
            const unsigned total = hlit + hdist;
            unsigned char *tmp = (unsigned char *)lut_ll;         // lens[0..19) is still in use: decode here, move afterwards
            while (n < total && err == B2_OK) {
                const unsigned d = bp >> 5, s = bp & 31;
                const u64 v = (((u64)b2_rfl(win[(d + 1) & 127]) << 32) | b2_rfl(win[d & 127])) >> s;   // >= 33 bits: two symbols at most 14 each
                const unsigned e = b2_rfl(lut_d[(unsigned)v & 127]);
                if (e == B2_ESC) {
                    err = B2_E_TABLE;
                    break;
                }
                const unsigned l = e & 15, sym = e >> 8;
                unsigned rep = 1, val = sym, used = l;
                if (sym == 16) {
                    if (n == 0) {
                        err = B2_E_TABLE;
                        break;
                    }
                    val = prev;
                    rep = 3 + ((unsigned)(v >> l) & 3);
                    used += 2;
                } else if (sym == 17) {
                    val = 0;
                    rep = 3 + ((unsigned)(v >> l) & 7);
                    used += 3;
                } else if (sym == 18) {
                    val = 0;
                    rep = 11 + ((unsigned)(v >> l) & 127);
                    used += 7;
                }
                if (n + rep > total) {
                    err = B2_E_TABLE;
                    break;
                }
                for (unsigned i = lane; i < rep; i += 64) tmp[n + i] = (unsigned char)val;
                n += rep;
                prev = val;
                bp += used;
                B2_ENSURE();
            }
            if (err != B2_OK) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
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
            if (b2_rfl(lens[256]) == 0) {
                err = B2_E_TABLE;
                break;
            }
        }
        if (!b2_build(lens, 288, B2_TB_LL, B2_MODE_LL, lut_ll, sorted_ll, meta_ll, lane) ||
            !b2_build(lens + 288, 32, B2_TB_D, B2_MODE_DIST, lut_d, sorted_d, meta_d, lane)) {
            err = B2_E_TABLE;
            break;
        }
        // ---- the symbols of this block, a window of 64 bit offsets at a time
        B2_MARK(11);                                                  // (block header + code lengths + the two table builds)
#if B2_W2
        // ---- TWO windows per trip: the lanes decode the symbols that would start at bp + lane AND at bp + 64 + lane, the chain is walked
        // through the first set and on through the second, and what a trip pays once whatever it decodes — the ring gather and the LUT round
        // trips, the prefix sum, the memory round trip of the own-lane copies, the cursor and the ring refill — is paid once per ~11 symbols
        // instead of once per 5.4 (a -DB2_WBITS=32 / 16 build prices that part at half of a 64-bit window's time: profiles/r06_ab_inflate_w2.txt).
        // A symbol's output size, literal byte and distance travel in one register (b2_spec), so the second set costs two live registers
        // across the first one's walk.
        for (;;) {
            unsigned pk0, sy0, pk1, sy1;
            b2_spec(win, lut_ll, lut_d, bp + (unsigned)lane, lane, pk0, sy0);
            b2_spec(win, lut_ll, lut_d, bp + 64u + (unsigned)lane, lane, pk1, sy1);
            u64 chain0 = 0, chain1 = 0;
            unsigned stop = 0, cur = 0, adv = 0;
            b2_walk(win, lut_ll, lut_d, sorted_ll, meta_ll, sorted_d, meta_d, bp, lane, pk0, sy0, chain0, cur, stop, err);
            if (err != B2_OK) break;
            if (stop != 2) {
                cur -= 64u;                                             // (64 .. 111 behind the first set's first bit)
                adv = 64u;
                b2_walk(win, lut_ll, lut_d, sorted_ll, meta_ll, sorted_d, meta_d, bp + 64u, lane, pk1, sy1, chain1, cur, stop, err);
                if (err != B2_OK) break;
            }
            adv += cur;
            unsigned ol0 = b2_sel(chain0, sy0 & 0x1ffu, 0u), ol1 = b2_sel(chain1, sy1 & 0x1ffu, 0u);
            unsigned incl0 = ol0, incl1 = ol1;
            b2_scan2(incl0, incl1);
            const unsigned tot0 = b2_rl(incl0, 63), tot = tot0 + b2_rl(incl1, 63);
            const unsigned pos0 = op + incl0 - ol0, pos1 = op + tot0 + incl1 - ol1;
            const bool copy0 = ol0 >= 3, copy1 = ol1 >= 3;
            const unsigned x0 = sy0 >> 9, x1 = sy1 >> 9;                // literal byte / distance - 1
            if (op + tot > isize || __ballot((copy0 && x0 >= pos0) || (copy1 && x1 >= pos1))) {
                err = op + tot > isize ? B2_E_OVERRUN : B2_E_DIST;
                break;
            }
            if (ol0 == 1) dst[pos0] = (unsigned char)x0;
            if (ol1 == 1) dst[pos1] = (unsigned char)x1;
            const unsigned srco0 = pos0 - x0 - 1u, srco1 = pos1 - x1 - 1u;
            // own-lane copies of BOTH sets: sources wholly before this trip's output (op) and at most B2_PARMAX bytes.  ALL their loads, ONE
            // wait, all their stores: a trip pays one memory round trip for them.  (The loaded words are deliberately not initialised and the
            // wait is unconditional: a zeroed register that a masked load may still be writing, or a wait the compiler can only see on one side
            // of an exec branch, makes its wait-count pass drain the queue — the previous trip's stores included — BEFORE the loads go out.)
            const bool wide0 = copy0 && srco0 + ol0 <= op && ol0 <= B2_PARMAX && ol0 >= 4 && srco0 + 16 <= isize;
            const bool wide1 = copy1 && srco1 + ol1 <= op && ol1 <= B2_PARMAX && ol1 >= 4 && srco1 + 16 <= isize;
            const bool tri0 = copy0 && ol0 == 3 && srco0 + 3 <= op, tri1 = copy1 && ol1 == 3 && srco1 + 3 <= op;   // (the dword read ends at srco + 4 <= pos + 1 <= isize)
            // The lanes on a chain are few (5.4 of a set's 64), so the two sets' own-lane copies share ONE set of copy instructions: a lane
            // copies the first set's match if it has one, else the second set's; a lane with one in both (one trip in eight) leaves its
            // second to the replay loop.
            const bool par0 = wide0 || tri0, par1 = (wide1 || tri1) && !par0;
            const bool wide = par0 ? wide0 : (wide1 && par1), tri = par0 ? tri0 : (tri1 && par1);
            const unsigned olm = par0 ? ol0 : ol1, posm = par0 ? pos0 : pos1, srcom = par0 ? srco0 : srco1;
            unsigned w3, tw;
            B2U128 vv;
            asm volatile("" : "=v"(w3), "=v"(tw), "=v"(vv.w[0]), "=v"(vv.w[1]), "=v"(vv.w[2]), "=v"(vv.w[3]));
#if B2_W2_EARLY
            // Each set's FIRST match that its lane cannot copy (longer than B2_PARMAX, or the lane's second) but that reads only what earlier
            // trips wrote is independent of everything this trip writes: ALL lanes copy it — lane k the k-th dword, the lane behind them the
            // last four bytes — with one load in the load phase and one store in the store phase, instead of a round trip of its own in
            // the replay loop (three replayed matches in four are of this kind: profiles/r04_inflate_windows.txt).
            const u64 ml0 = __ballot(copy0 && !par0 && srco0 + ol0 <= op && ol0 - 4u <= 252u);
            const u64 ml1 = __ballot(copy1 && !par1 && srco1 + ol1 <= op && ol1 - 4u <= 252u);
            unsigned e_len0 = 0, e_p0 = 0, e_so0 = 0, e_len1 = 0, e_p1 = 0, e_so1 = 0;
            u64 e_bit0 = 0, e_bit1 = 0;
            if (ml0) {
                const unsigned l = (unsigned)__builtin_ctzll(ml0);
                e_bit0 = 1ull << l;
                e_len0 = b2_rl(ol0, l), e_p0 = b2_rl(pos0, l), e_so0 = b2_rl(srco0, l);
            }
            if (ml1) {
                const unsigned l = (unsigned)__builtin_ctzll(ml1);
                e_bit1 = 1ull << l;
                e_len1 = b2_rl(ol1, l), e_p1 = b2_rl(pos1, l), e_so1 = b2_rl(srco1, l);
            }
            const unsigned nd0 = e_len0 >> 2, nd1 = e_len1 >> 2;
            const bool ea0 = (unsigned)lane < nd0 || ((unsigned)lane == nd0 && (e_len0 & 3u));
            const bool ea1 = (unsigned)lane < nd1 || ((unsigned)lane == nd1 && (e_len1 & 3u));
            const unsigned eo0 = (unsigned)lane < nd0 ? 4u * (unsigned)lane : e_len0 - 4u, eo1 = (unsigned)lane < nd1 ? 4u * (unsigned)lane : e_len1 - 4u;
            unsigned ev0, ev1;
            asm volatile("" : "=v"(ev0), "=v"(ev1));
            if (ea0) ev0 = reinterpret_cast<const B2U32 *>(dst + (e_so0 + eo0))->v;
            if (ea1) ev1 = reinterpret_cast<const B2U32 *>(dst + (e_so1 + eo1))->v;
#endif
            if (tri) w3 = reinterpret_cast<const B2U32 *>(dst + srcom)->v;
            if (wide) {
                vv = *reinterpret_cast<const B2U128 *>(dst + srcom);
                tw = reinterpret_cast<const B2U32 *>(dst + (srcom + olm - 4u))->v;
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0), nothing else
#if B2_W2_EARLY
            if (ea0) reinterpret_cast<B2U32 *>(dst + (e_p0 + eo0))->v = ev0;
            if (ea1) reinterpret_cast<B2U32 *>(dst + (e_p1 + eo1))->v = ev1;
#endif
            if (tri) {
                reinterpret_cast<B2U16 *>(dst + posm)->v = (unsigned short)w3;
                dst[posm + 2] = (unsigned char)(w3 >> 16);
            }
            if (wide) {
                const bool g8 = olm >= 8, g12 = olm >= 12, g16 = olm >= 16;
                reinterpret_cast<B2U32 *>(dst + posm)->v = vv.w[0];
                reinterpret_cast<B2U32 *>(dst + (posm + (g8 ? 4u : 0u)))->v = g8 ? vv.w[1] : vv.w[0];
                reinterpret_cast<B2U32 *>(dst + (posm + (g12 ? 8u : 0u)))->v = g12 ? vv.w[2] : vv.w[0];
                reinterpret_cast<B2U32 *>(dst + (posm + (g16 ? 12u : 0u)))->v = g16 ? vv.w[3] : vv.w[0];
                reinterpret_cast<B2U32 *>(dst + (posm + olm - 4u))->v = tw;
            }
            // the others in stream order (they may read each other's output, and the second set's may read the first set's)
#define B2_REPLAY(copy_, par_, ol_, x_, pos_, srco_, done_)                                                            \
            do {                                                                                                       \
                u64 mm = __ballot(copy_ && !par_) & ~(done_);                                                          \
                while (mm) {                                                                                           \
                    const unsigned l = (unsigned)__builtin_ctzll(mm);                                                  \
                    mm &= ~(1ull << l);                                                                                \
                    const unsigned len = b2_rl(ol_, l), dd = b2_rl(x_, l) + 1u, p = b2_rl(pos_, l), so = b2_rl(srco_, l); \
                    if (dd >= len) {                                                                                   \
                        for (unsigned i = (unsigned)lane; i < len; i += 64) dst[p + i] = dst[so + i];                  \
                    } else {                                                                                           \
                        unsigned j = (unsigned)lane % dd;                                                              \
                        const unsigned step = 64u % dd;                                                                \
                        for (unsigned i = (unsigned)lane; i < len; i += 64) {                                          \
                            dst[p + i] = dst[so + j];                                                                  \
                            j += step;                                                                                 \
                            j -= j >= dd ? dd : 0u;                                                                    \
                        }                                                                                              \
                    }                                                                                                  \
                }                                                                                                      \
            } while (0)
#if B2_W2_EARLY
            B2_REPLAY(copy0, par0, ol0, x0, pos0, srco0, e_bit0);
            B2_REPLAY(copy1, par1, ol1, x1, pos1, srco1, e_bit1);
#else
            B2_REPLAY(copy0, par0, ol0, x0, pos0, srco0, 0ull);
            B2_REPLAY(copy1, par1, ol1, x1, pos1, srco1, 0ull);
#endif
#undef B2_REPLAY
            op += tot;
            bp += adv;
            if (bp > end_bit) {                                     // a valid block ends (EOB included) inside the payload
                err = B2_E_INPUT;
                break;
            }
            B2_ENSURE();
            if (err != B2_OK || stop == 2) break;
        }
#else
        for (;;) {
            // every lane: the symbol that would start at bit bp + lane
            // (three dwords from the ring; every field but the distance's extra bits lies in the first 32 stream bits:
            //  9-bit code + 5 extra + 8-bit code = 22, so 64-bit shifts are not needed)
            B2_MARK(7);                                               // (whatever came before this window: tables, headers, the previous window's tail)
            const unsigned q = bp + (unsigned)lane, qd = (q >> 5) & 127, qs = q & 31;
            const unsigned wa = win[qd], wb = win[qd + 1], wc = win[qd + 2];      // one address, three reads (the ring's mirror: no wrap)
            const unsigned lo = __builtin_amdgcn_alignbit(wb, wa, qs), hi = __builtin_amdgcn_alignbit(wc, wb, qs);
            const unsigned e1 = lut_ll[lo & ((1u << B2_TB_LL) - 1)];
            const unsigned step1 = B2_STEP(e1);                     // literal: l1; length: l1 + eb1 (<= 9 + 5); 256 + l1 at the end of block; 128 = escape
            const unsigned e2 = lut_d[__builtin_amdgcn_ubfe(lo, step1, B2_TB_D)];   // (offset = step1 & 31: only a length code's lane uses e2)
            const bool is_len = e1 & B2_F_LEN;
            // next[i] = i + bits consumed; a chain ends at a value >= 64: [128, 256) = a code the tables do not resolve (the literal /
            // length code itself or the distance code behind it), >= 256 = end of block (256 + the bit after it)
            const unsigned nxt = (unsigned)lane + step1 + (is_len ? B2_STEP(e2) : 0u);
#if B2_HOP2
            // TWO symbols per hop of the chain walk: every lane also learns where the symbol BEHIND its own would end (one LDS permute:
            // lane i reads next[next[i]]), and the walk reads both with one v_readlane — the scalar / vector hand-over, half of a hop's
            // ≈ 185 cycles, is paid once per two symbols.  m1 = the lane one symbol ahead, 64 when that hop leaves the window;
            // n2 = where the walk stands after two symbols (the exit value of whichever hop leaves the window first).
            const unsigned m1 = nxt < (unsigned)B2_WBITS ? nxt : 64u;
            const unsigned n2r = (unsigned)__builtin_amdgcn_ds_bpermute((int)(m1 << 2), (int)nxt);   // (m1 = 64 reads lane 0: not used)
            const unsigned pack2 = m1 | ((nxt < (unsigned)B2_WBITS ? n2r : nxt) << 8);
#endif
#ifdef B2_PROF
            asm volatile("" :: "v"(nxt));
            B2_MARK(0);
#endif
            // what a symbol starting at this lane's bit would write — 1 byte (literal), its length (match), nothing (end of block) — and,
            // for a match, from how far back
            const unsigned l1 = e1 & 15, eb1 = (e1 >> 4) & 15;
            unsigned litv = (e1 >> 8) & 0x1ff;                      // literal byte / base length
            unsigned mlen = litv + __builtin_amdgcn_ubfe(lo >> l1, 0u, eb1);
            unsigned olraw = is_len ? mlen : ((e1 & B2_F_EOB) ? 0u : 1u);
            const unsigned l2 = e2 & 15, eb2 = (e2 >> 4) & 15, base2 = (e2 >> 8) & 0x7fff;
            unsigned dist = base2 + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi, lo, (step1 + l2) & 31), 0u, eb2);
#ifdef B2_PROF
            asm volatile("" :: "v"(dist), "v"(olraw));
            B2_MARK(1);
#endif
            // The true chain of symbol starts (scalar: one readlane per symbol).  A code longer than the LUTs (one symbol in ten on
            // BAM-shaped data) does not end the window: it is resolved right here by the scalar canonical walk, its lane is patched with
            // what the symbol writes, and the chain goes on behind it — the symbol's bytes leave with the window's other output instead of
            // costing an iteration and a memory round trip of their own.
            unsigned cur = 0;
            u64 chain = 0;
            unsigned stop = 0;                                      // 2 = end of block
            for (;;) {
                while (cur < (unsigned)B2_WBITS) {
#if B2_HOP2
                    const unsigned p2 = b2_rl(pack2, cur);
                    asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(cur));
                    asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(p2));       // bits 5:0 = m1 & 63: the next symbol's lane, or lane 0 — always on the
                    cur = p2 >> 8;                                             // chain — when that hop leaves the window
#else
                    asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(cur));      // chain |= 1 << cur in ONE scalar instruction (the scalar unit is shared by the CU's 32 waves)
                    cur = b2_rl(nxt, cur);
#endif
                }
#ifdef B2_PROF
                B2_MARK(9);                                         // (the hops alone; what is left under mark 2 is the long-code path)
                if (cur >= 128 && cur < 256) pf[10]++;
#endif
                // (the common exit first — the walk left the window, 64 <= cur < 128: one compare and one branch on the path every window takes;
                //  tested behind the end-of-block case it cost 1.5 % of the launch, profiles/r06_ab_inflate_w2.txt (10))
                if (__builtin_expect(cur < 128, 1)) break;
                if (cur >= 256) {
                    stop = 2;
                    cur -= 256;
                    break;
                }
                const unsigned at = 63u - (unsigned)__builtin_clzll(chain);      // the chain's last member is the symbol the LUTs did not resolve
                const unsigned qq = bp + at, d = (qq >> 5) & 127, sh = qq & 31;
                const unsigned w0 = b2_rfl(win[d]), w1 = b2_rfl(win[d + 1]), w2 = b2_rfl(win[d + 2]);
                u64 v = ((u64)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
                unsigned e = b2_rfl(lut_ll[(unsigned)v & ((1u << B2_TB_LL) - 1)]);
                if (e == B2_ESC) e = b2_long_code((unsigned)v, B2_TB_LL, B2_MODE_LL, sorted_ll, meta_ll, lane);
                if (e == B2_ESC) {
                    err = B2_E_SYMBOL;
                    break;
                }
                unsigned used = e & 15;
                v >>= used;
                unsigned s_ol = 1, s_len = 0, s_dist = 0;
                if (e & B2_F_EOB) {
                    s_ol = 0;
                    stop = 2;
                } else if (e & B2_F_LEN) {
                    unsigned eb = (e >> 4) & 15;
                    s_len = ((e >> 8) & 0x1ff) + ((unsigned)v & ((1u << eb) - 1));
                    v >>= eb;
                    used += eb;
                    unsigned ed = b2_rfl(lut_d[(unsigned)v & ((1u << B2_TB_D) - 1)]);
                    if (ed == B2_ESC) ed = b2_long_code((unsigned)v, B2_TB_D, B2_MODE_DIST, sorted_d, meta_d, lane);
                    if (ed == B2_ESC) {
                        err = B2_E_DIST;
                        break;
                    }
                    v >>= ed & 15;
                    eb = (ed >> 4) & 15;
                    s_dist = ((ed >> 8) & 0x7fff) + ((unsigned)v & ((1u << eb) - 1));
                    used += (ed & 15) + eb;
                    s_ol = s_len;
                }
                b2_wl(olraw, s_ol, at);
                b2_wl(mlen, s_len, at);
                b2_wl(dist, s_dist, at);
                b2_wl(litv, (e >> 8) & 0xff, at);
                cur = at + used;                                    // (<= 63 + 48: the ring holds bp .. bp + 160 bits)
                if (stop == 2) break;
            }
            if (err != B2_OK) break;
            B2_MARK(2);
            const unsigned ol = b2_sel(chain, olraw, 0u);           // off the chain: nothing
            const unsigned incl = b2_scan(ol);
            const unsigned tot = b2_rl(incl, 63);
            const unsigned pos = op + incl - ol;
            const bool copy = ol >= 3;                              // (a match is at least 3 bytes long, a literal exactly 1)
            if (op + tot > isize || __ballot(copy && dist > pos)) {
                err = op + tot > isize ? B2_E_OVERRUN : B2_E_DIST;
                break;
            }
            B2_MARK(3);
#if !defined(B2_EXP_NOLIT) && !(B2_PHASED && !B2_PIPE)   // B2_EXP_*: ablation switches (tools/build_variant.sh) behind the stage costs quoted in DESIGN.md 3.6
            if (ol == 1) dst[pos] = (unsigned char)litv;
#endif
            const unsigned srco = pos - dist;                       // first source byte of this lane's match
            // Matches whose source lies wholly before this window's output cannot depend on anything decoded in it: each of
            // those is copied by its own lane, all at once (3 bytes unconditionally — the minimum match — then the rest).
#ifdef B2_EXP_NOOWN    // ablation (output wrong): what the own-lane copies cost
            const bool par = false;
#else
            const bool par = copy && srco + mlen <= op && mlen <= B2_PARMAX;
#endif
            const bool wide = par && mlen >= 4 && srco + 16 <= isize;   // the 16-byte read stays inside this block's output
#if B2_PIPE
            // (measurement variant) The copies are PIPELINED over two windows: this window was decoded while the previous window's loads
            // were in flight; now the previous window's bytes are stored (B2_FLUSH waits for them here, one decode later than it used to), and
            // only then are this window's loads issued — a source may lie in what the flush has just written, and the memory operations
            // of one wave stay in order.
            B2_FLUSH();
            pd_kind = wide ? 1u : par ? 2u : copy ? 3u : 0u;
            pd_pos = pos;
            pd_mlen = mlen;
            pd_dist = dist;
            if (wide) {
                const unsigned char *const s_ = dst + srco;
                pd_v = *reinterpret_cast<const B2U128 *>(s_);
                pd_tail = reinterpret_cast<const B2U32 *>(s_ + mlen - 4)->v;
            }
#else
#if B2_PHASED
            // Every copy whose source lies before this window's output is independent of the window's other copies, so ALL their loads are
            // issued first and all their stores afterwards: the byte path (3-byte matches, block ends), the wide path, and the window's
            // first match that is too long for its own lane but reads only earlier windows' output (three replayed matches in four) — ONE
            // memory round trip where the three paths took one each, one after the other.  (The order matters for a second reason: a
            // wave's memory operations return in order, so a wait for a load that has younger STORES behind it in the queue — the
            // compiler cannot count them across exec regions and waits for everything — waits for their acknowledgements too, which is
            // what made "loads early, stores late" for one path alone slower, twice.)
            const bool bytep = par && !wide;
            unsigned char by0 = 0, by1 = 0, by2 = 0;
            B2U128 v = {{0, 0, 0, 0}};
            unsigned tailw = 0;
            unsigned e_len = 0, e_p = 0, e_lane = 0;
            unsigned char e_v = 0;
            if (bytep) {
                by0 = dst[srco];
                by1 = dst[srco + 1];
                by2 = dst[srco + 2];
            }
            if (wide) {
                v = *reinterpret_cast<const B2U128 *>(dst + srco);
                tailw = reinterpret_cast<const B2U32 *>(dst + (srco + mlen - 4u))->v;
            }
            {
                const u64 mi = B2_EARLYM ? __ballot(copy && !par && srco + mlen <= op && mlen <= 64u) : 0ull;
                if (mi) {
                    e_lane = (unsigned)__builtin_ctzll(mi);
                    e_len = b2_rl(mlen, e_lane);
                    e_p = b2_rl(pos, e_lane);
                    const unsigned so = b2_rl(srco, e_lane);
                    if ((unsigned)lane < e_len) e_v = dst[so + (unsigned)lane];
                }
            }
            // every load of the phase is back before the first store goes out (the builtin, not inline asm: the compiler's own wait-count
            // pass sees it and places no further waits between the stores — placed by itself, each path's wait for ITS loads counts the
            // stores of the path in front of it as well)
            __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0), nothing else; unconditional, or the pass cannot rely on it
            if (ol == 1) dst[pos] = (unsigned char)litv;            // (the literals too go out behind the wait, not in front of the loads)
            if (e_len && (unsigned)lane < e_len) dst[e_p + (unsigned)lane] = e_v;
            if (bytep) {
                dst[pos] = by0;
                dst[pos + 1] = by1;
                dst[pos + 2] = by2;
                for (unsigned k = 3; k < mlen; k++) dst[pos + k] = dst[srco + k];
            }
            if (wide) {
                const bool g8 = mlen >= 8, g12 = mlen >= 12, g16 = mlen >= 16;
                reinterpret_cast<B2U32 *>(dst + pos)->v = v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g8 ? 4u : 0u)))->v = g8 ? v.w[1] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g12 ? 8u : 0u)))->v = g12 ? v.w[2] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g16 ? 12u : 0u)))->v = g16 ? v.w[3] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + mlen - 4u))->v = tailw;
            }
#else
            // (a 3-byte match, or a source in the block's last bytes.  Its source lies before this window's output, so the three bytes every
            //  match has are LOADED FIRST and stored together: one memory round trip; written as a plain byte loop the compiler, which
            //  cannot see that, waited for every byte before storing it — three round trips for the commonest match of zlib level 1)
            if (par && !wide) {
                // (ONE dword load — srco + 4 <= pos + 1 <= isize: the match's own three output bytes lie behind its source — and a 16-bit +
                //  an 8-bit store: three memory instructions instead of six)
                const unsigned w3 = reinterpret_cast<const B2U32 *>(dst + srco)->v;
                reinterpret_cast<B2U16 *>(dst + pos)->v = (unsigned short)w3;
                dst[pos + 2] = (unsigned char)(w3 >> 16);
                for (unsigned k = 3; k < mlen; k++) dst[pos + k] = dst[srco + k];
            }
            if (wide) {
                // (every address is the block's base in a scalar pair + a 32-bit index in a vector register: formed as pointers first,
                //  each store cost a 64-bit add and a zeroed high half on the vector unit, which is the busiest unit of this kernel)
                const B2U128 v = *reinterpret_cast<const B2U128 *>(dst + srco);
                const unsigned tailw = reinterpret_cast<const B2U32 *>(dst + (srco + mlen - 4u))->v;
                // every store unconditional: a dword the match does not reach is written as dword 0 once more, the tail always (it equals the
                // last whole dword when the length is a multiple of four) — two selects per store instead of an exec-mask branch (three
                // scalar instructions each)
                const bool g8 = mlen >= 8, g12 = mlen >= 12, g16 = mlen >= 16;
                reinterpret_cast<B2U32 *>(dst + pos)->v = v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g8 ? 4u : 0u)))->v = g8 ? v.w[1] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g12 ? 8u : 0u)))->v = g12 ? v.w[2] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + (g16 ? 12u : 0u)))->v = g16 ? v.w[3] : v.w[0];
                reinterpret_cast<B2U32 *>(dst + (pos + mlen - 4u))->v = tailw;
            }
#endif
#endif
#ifdef B2_STATS
            {
                const u64 m_lit = __ballot(ol == 1), m_par = __ballot(par), m_long = __ballot(copy && !par && srco + mlen <= op), m_dep = __ballot(copy && !par && srco + mlen > op);
                unsigned mb = copy ? mlen : 0u, rb = (copy && !par) ? mlen : 0u;
                for (int d_ = 32; d_ > 0; d_ >>= 1) {
                    mb += __shfl_xor(mb, d_);
                    rb += __shfl_xor(rb, d_);
                }
                if (lane == 0) {
                    atomicAdd(&b2_stats[0], 1ull);
                    atomicAdd(&b2_stats[1], (unsigned long long)__popcll(chain));
                    atomicAdd(&b2_stats[2], (unsigned long long)__popcll(m_lit));
                    atomicAdd(&b2_stats[3], (unsigned long long)__popcll(m_par));
                    atomicAdd(&b2_stats[4], (unsigned long long)__popcll(m_long));
                    atomicAdd(&b2_stats[5], (unsigned long long)__popcll(m_dep));
                    atomicAdd(&b2_stats[6], 0ull);
                    atomicAdd(&b2_stats[7], (unsigned long long)mb);
                    atomicAdd(&b2_stats[8], (unsigned long long)rb);
                }
            }
#endif
#ifdef B2_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the copies' stores are issued: charge their completion to the copies)
            B2_MARK(4);
#endif
#if !B2_PIPE
            u64 mm = __ballot(copy && !par);
#ifdef B2_EXP_NOREPLAY  // ablation (output wrong): what the replayed matches cost
            mm = 0;
#endif
#if B2_PHASED
            if (e_len) mm &= ~(1ull << e_lane);
#endif
            while (mm) {                                            // the others in stream order (they may read each other's output)
                const unsigned l = (unsigned)__builtin_ctzll(mm);
                mm &= ~(1ull << l);
                const unsigned len = b2_rl(mlen, l), dd = b2_rl(dist, l), p = b2_rl(pos, l), so = b2_rl(srco, l);
                if (dd >= len) {                                    // (uniform) source and destination do not overlap: a plain copy —
                    for (unsigned i = (unsigned)lane; i < len; i += 64) dst[p + i] = dst[so + i];      // no per-lane division in the common case
                } else {                                            // a distance shorter than the match repeats its source: byte i comes from i mod dd
                    unsigned j = (unsigned)lane % dd;
                    const unsigned step = 64u % dd;
                    for (unsigned i = (unsigned)lane; i < len; i += 64) {
                        dst[p + i] = dst[so + j];
                        j += step;
                        j -= j >= dd ? dd : 0u;
                    }
                }
            }
#endif
            if (err != B2_OK) break;
#ifdef B2_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            B2_MARK(5);
            pf[8]++;
#endif
            op += tot;
            bp += cur;
            if (bp > end_bit) {                                     // a valid block ends (EOB included) inside the payload
                err = B2_E_INPUT;
                break;
            }
            B2_ENSURE();
            B2_MARK(6);
            if (err != B2_OK || stop == 2) break;
        }
#endif
#if B2_PIPE
        if (err == B2_OK) B2_FLUSH();                              // the last window's copies, before the next DEFLATE block writes behind them
#endif
    }
    if (err == B2_OK && op != isize) err = B2_E_SIZE;
    if (err == B2_OK && bp > end_bit + 7) err = B2_E_INPUT;
    if (lane == 0) status[b] = err;
#ifdef B2_PROF
    B2_MARK(7);
    if (lane == 0)
        for (int k_ = 0; k_ < 12; k_++) atomicAdd(&b2_prof[k_], (unsigned long long)pf[k_]);
#endif
    __builtin_amdgcn_wave_barrier();                          // (the next block's staging writes the ring this one may still be reading)
    }
#undef B2_ENSURE
#undef B2_STAGE
#if B2_PIPE
#undef B2_FLUSH
#endif
}

void tdt_bz_launch_lanes(hipStream_t st, int num_cu, int reserve, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks,
                         unsigned char *d_out, unsigned *d_status, unsigned *d_next_block) {
#ifdef B2_NOPERSIST
    const size_t resident = ~(size_t)0;
#else
    const int per_cu = 4 * B2_OCC / B2_WAVES;                                          // workgroups a CU holds at 8 waves per SIMD
    const size_t resident = (size_t)num_cu * (size_t)(reserve > 0 && reserve < per_cu ? per_cu - reserve : per_cu);
#endif
    const unsigned grid = (unsigned)std::min(resident, (nblocks + B2_WAVES - 1) / B2_WAVES);
    hipLaunchKernelGGL(bgzf_inflate_lanes, dim3(grid), dim3(64 * B2_WAVES), 0, st, d_comp, d_blocks, (int)nblocks, d_out, d_status, d_next_block);
}
