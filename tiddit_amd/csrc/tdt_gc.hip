// Binned GC / N-mask histogram for gfx950 (MI355X).
//
// Replaces the per-character Python loop of tiddit_gc.binned_gc (tiddit_gc.pyx:14-31):
//   n = #{N,n}, gc = #{C,c,G,g}, chars = bytes in the bin (the contig's last bin may be short)
//   out = -1 if n/bin_size > n_cutoff else round_half_even(100*gc/chars)
// HBM-bound byte stream (1 B/base in, 1 B/bin out).  Small bins (the reference always uses 50):
//   phase 1  every lane loads 16 coalesced bytes, classifies them with SWAR compares into a 16-bit
//            GC mask and a 16-bit N mask and drops both into LDS bit arrays (one bit per base);
//   phase 2  every lane owns one bin = one bit range of the LDS arrays and popcounts it (<= 3 words
//            for 50-bp bins), then rounds in integers and stores one int8.
// Bins wider than GC_SMALL_MAX use one workgroup per bin with a wavefront/LDS reduction.
#include "tdt_common.h"

#define GC_THREADS 256
#define GC_SMALL_MAX 2048

// Marked word of "byte == K" for a 7-bit constant K, per byte 0x7f on a match and 0xff elsewhere: with w7 = the byte's
// low seven bits (case bit forced on), (w7 ^ K) + 0x7f has bit 7 set iff the low bits differ and never carries out of its
// byte; OR-ing the raw word in passes a set bit 7 of the input through (such a byte cannot match) and 0x7f fills the rest.
// Three-operand VALU ops make this two instructions per test: v_xad_u32, v_or3_b32.
__device__ __forceinline__ unsigned gc_mark_eq(unsigned w7, unsigned k, unsigned w) {
    return ((w7 ^ k) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu;
}
// 16-bit match mask (bit i = byte i of the 16) from the four marked words: v_dot4_u32_u8 weighs byte j of a word with
// 1,2,4,8 (words 0 and 2) or 16,32,64,128 (words 1 and 3) and adds; a marked byte is 0xff - 0x80*match, so two words sum
// to 0xff*255 - 0x80*M8 with M8 their 8-bit match mask.  The two halves are joined BEFORE the subtraction
// (0x80*M8 < 2^15, so the halves cannot run into each other): M16 = (65025*257 - (d01 + d23*256)) >> 7.
__device__ __forceinline__ unsigned gc_mask16(unsigned z0, unsigned z1, unsigned z2, unsigned z3) {
    const unsigned d01 = __builtin_amdgcn_udot4(z0, 0x08040201u, __builtin_amdgcn_udot4(z1, 0x80402010u, 0u, false), false);
    const unsigned d23 = __builtin_amdgcn_udot4(z2, 0x08040201u, __builtin_amdgcn_udot4(z3, 0x80402010u, 0u, false), false);
    return (65025u * 257u - ((d23 << 8) + d01)) >> 7;
}

// Fold case ('C'/'c' -> 0x63, 'G'/'g' -> 0x67, 'N'/'n' -> 0x6e); 'c' and 'g' differ in bit 2 only, so ONE test with bit 2
// dropped finds both.  Six VALU instructions per word for both tests; integer multiplies are quarter rate on CDNA, so the
// bit gather runs on the dot unit instead (gc_mask16).
__device__ __forceinline__ void gc_classify16(const uint4 v, unsigned &gc16, unsigned &n16) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned zg[4], zn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned w7 = (w[k] & 0x7f7f7f7fu) | 0x20202020u;      // v_and_or_b32: low seven bits, lower case
        const unsigned w7g = (w[k] & 0x7b7b7b7bu) | 0x20202020u;     // ... and bit 2 dropped: 'c' and 'g' both become 0x63
        zg[k] = gc_mark_eq(w7g, 0x63636363u, w[k]);
        zn[k] = gc_mark_eq(w7, 0x6e6e6e6eu, w[k]);
    }
    gc16 = gc_mask16(zg[0], zg[1], zg[2], zg[3]);
    n16 = gc_mask16(zn[0], zn[1], zn[2], zn[3]);
}

// 16 bytes at byte offset g (16-aligned); bytes at or beyond len read as 0
__device__ __forceinline__ uint4 gc_load16(const uint8_t *seq, long long g, long long len) {
    if (g + 16 <= len) return *reinterpret_cast<const uint4 *>(seq + g);
    unsigned w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++)
        if (g + i < len) w[i >> 2] |= (unsigned)seq[g + i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ signed char gc_result(unsigned long long gc, unsigned long long n, unsigned long long chars,
                                                 unsigned long long n_min) {
    if (n >= n_min) return -1;                       // n/bin_size > n_cutoff   (tiddit_gc.pyx:27)
    const unsigned long long a = 100ull * gc;        // round(100*gc/chars), Python half-to-even (:30)
    unsigned long long q = a / chars;
    const unsigned long long r2 = 2ull * (a - q * chars);
    if (r2 > chars || (r2 == chars && (q & 1ull))) q++;
    return (signed char)q;
}

// the same in 32 bits for the small-bin kernel (gc, n, chars <= 2048): a 64-bit division is ~150 VALU instructions in
// software, and with five bins per lane it was half of that kernel's vector work
__device__ __forceinline__ signed char gc_result32(unsigned gc, unsigned n, unsigned chars, unsigned n_min) {
    if (n >= n_min) return -1;
    const unsigned a = 100u * gc;
    unsigned q = a / chars;
    const unsigned r2 = 2u * (a - q * chars);
    if (r2 > chars || (r2 == chars && (q & 1u))) q++;
    return (signed char)q;
}

__global__ __launch_bounds__(GC_THREADS) void gc_small_bins(const uint8_t *__restrict__ seq, long long len, int bin_size,
                                                            int tile_bins, long long nbins, unsigned long long n_min,
                                                            signed char *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned gc_smem[];
    const int tile_bytes = tile_bins * bin_size;  // multiple of 16
    const int chunks = tile_bytes >> 4;
    const int words = (chunks + 1) >> 1;          // u32 words per bit array
    unsigned *gbits = gc_smem;
    unsigned *nbits = gc_smem + words;
    unsigned short *g16 = reinterpret_cast<unsigned short *>(gbits);
    unsigned short *n16 = reinterpret_cast<unsigned short *>(nbits);
    const int tid = threadIdx.x;

    const unsigned nmin32 = n_min > 0xffffffffull ? 0xffffffffu : (unsigned)n_min;
    for (long long tile = blockIdx.x; tile * tile_bins < nbins; tile += gridDim.x) {
        const long long T0 = tile * (long long)tile_bytes;
        // A tile that lies wholly inside the sequence (all but the last) is addressed as uniform base + 32-bit lane
        // offset and has only full bins; the 64-bit bounds arithmetic is left to the one tile at the end.
        const bool whole = T0 + tile_bytes <= len && (tile + 1) * tile_bins <= nbins;
        const uint8_t *const tp = seq + T0;
        // phase 1, four chunks per lane per trip: the four 16-byte loads are issued back to back
        // (prefetching the next trip across the popcount phase was measured slower: 0.63 vs 0.57 ms per 3 Gb)
        for (int c0 = 0; c0 < chunks; c0 += 4 * GC_THREADS) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = c0 + k * GC_THREADS + tid;
                if (c < chunks) v[k] = whole ? *reinterpret_cast<const uint4 *>(tp + 16u * (unsigned)c) : gc_load16(seq, T0 + 16ll * c, len);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = c0 + k * GC_THREADS + tid;
                if (c < chunks) {
                    unsigned gm, nm;
                    gc_classify16(v[k], gm, nm);
                    g16[c] = (unsigned short)gm;
                    n16[c] = (unsigned short)nm;
                }
            }
        }
        if (tid == 0 && (chunks & 1)) {  // zero the unused upper half of the last word
            g16[chunks] = 0;
            n16[chunks] = 0;
        }
        __syncthreads();
        signed char *const to = out + tile * tile_bins;
        for (int j = tid; j < tile_bins; j += GC_THREADS) {
            if (!whole && tile * tile_bins + j >= nbins) break;
            const int lo = j * bin_size;
            int chars = bin_size;
            if (!whole) {
                const long long rem = len - (T0 + lo);
                chars = rem < bin_size ? (int)rem : bin_size;
            }
            const int hi = lo + chars - 1;  // inclusive last bit
            const int w0 = lo >> 5, w1 = hi >> 5;
            unsigned gcnt = 0, ncnt = 0;
            for (int w = w0; w <= w1; w++) {
                unsigned m = 0xffffffffu;
                if (w == w0) m &= 0xffffffffu << (lo & 31);
                if (w == w1) m &= 0xffffffffu >> (31 - (hi & 31));
                gcnt += __popc(gbits[w] & m);
                ncnt += __popc(nbits[w] & m);
            }
            to[j] = gc_result32(gcnt, ncnt, (unsigned)chars, nmin32);
        }
        __syncthreads();
    }
}

// The same histogram straight from the FASTA bytes of a contig (line feeds still in place): base b sits at byte
// fb(b) = (b / linebases) * linewidth + b % linebases of the contig's byte range.  Line-end bytes classify as neither GC nor
// N, so phase 1 is unchanged; phase 2 gives bin j the bit range [fb(j*bin), fb(end-1)] and its true base count.  The
// reference reads the file through pysam.FastaFile one 50-bp slice at a time (tiddit_gc.pyx:14-19); this removes the
// host-side newline stripping (two full copies of a 3 GB genome) from the path.
__global__ __launch_bounds__(GC_THREADS) void gc_fasta_bins(const uint8_t *__restrict__ raw, long long nbytes, long long len, int bin_size,
                                                            unsigned linebases, unsigned linewidth, unsigned magic, int shift, int tile_bins,
                                                            int max_chunks, long long nbins, unsigned n_min, signed char *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned gc_smem[];
    const int words = (max_chunks + 1) >> 1;
    unsigned *gbits = gc_smem;
    unsigned *nbits = gc_smem + words;
    unsigned short *g16 = reinterpret_cast<unsigned short *>(gbits);
    unsigned short *n16 = reinterpret_cast<unsigned short *>(nbits);
    const int tid = threadIdx.x;
    auto fb = [&](unsigned b) -> long long {          // byte offset of base b (b < 2^31)
        const unsigned line = shift < 0 ? b : (unsigned)(__umulhi(b, magic) >> shift);
        return (long long)line * linewidth + (b - line * linebases);
    };
    for (long long tile = blockIdx.x; tile * tile_bins < nbins; tile += gridDim.x) {
        const long long B0 = tile * (long long)tile_bins * bin_size;
        long long B1 = B0 + (long long)tile_bins * bin_size;
        if (B1 > len) B1 = len;
        const long long f0 = fb((unsigned)B0) & ~15ll;                 // 16-byte aligned start of the tile's bytes
        const long long f1 = fb((unsigned)(B1 - 1)) + 1;               // one past its last base
        const int chunks = (int)((f1 - f0 + 15) >> 4);                 // <= max_chunks by construction
        for (int c0 = 0; c0 < chunks; c0 += 4 * GC_THREADS) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = c0 + k * GC_THREADS + tid;
                if (c < chunks) v[k] = gc_load16(raw, f0 + 16ll * c, nbytes);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = c0 + k * GC_THREADS + tid;
                if (c < chunks) {
                    unsigned gm, nm;
                    gc_classify16(v[k], gm, nm);
                    g16[c] = (unsigned short)gm;
                    n16[c] = (unsigned short)nm;
                }
            }
        }
        if (tid == 0 && (chunks & 1)) {
            g16[chunks] = 0;
            n16[chunks] = 0;
        }
        __syncthreads();
        for (int j = tid; j < tile_bins; j += GC_THREADS) {
            const long long bin = tile * tile_bins + j;
            if (bin >= nbins) break;
            const long long b0 = bin * bin_size;
            const long long b1 = b0 + bin_size < len ? b0 + bin_size : len;
            const int lo = (int)(fb((unsigned)b0) - f0), hi = (int)(fb((unsigned)(b1 - 1)) - f0);   // inclusive bit range
            const int w0 = lo >> 5, w1 = hi >> 5;
            unsigned gcnt = 0, ncnt = 0;
            for (int w = w0; w <= w1; w++) {
                unsigned m = 0xffffffffu;
                if (w == w0) m &= 0xffffffffu << (lo & 31);
                if (w == w1) m &= 0xffffffffu >> (31 - (hi & 31));
                gcnt += __popc(gbits[w] & m);
                ncnt += __popc(nbits[w] & m);
            }
            out[bin] = gc_result32(gcnt, ncnt, (unsigned)(b1 - b0), n_min);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(GC_THREADS) void gc_large_bins(const uint8_t *__restrict__ seq, long long len,
                                                            long long bin_size, long long nbins, unsigned long long n_min,
                                                            signed char *__restrict__ out) {
    __shared__ unsigned long long red[2 * (GC_THREADS / 64)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (long long bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
        const long long lo = bin * bin_size;
        const long long hi = min(len, lo + bin_size);  // exclusive
        const long long a0 = lo & ~15ll;
        unsigned long long gcnt = 0, ncnt = 0;
        for (long long g = a0 + 16ll * tid; g < hi; g += 16ll * GC_THREADS) {
            unsigned gm, nm;
            gc_classify16(gc_load16(seq, g, len), gm, nm);
            unsigned valid = 0xffffu;
            if (g < lo) valid &= 0xffffu << (lo - g);
            if (g + 16 > hi) valid &= 0xffffu >> (g + 16 - hi);
            gcnt += __popc(gm & valid);
            ncnt += __popc(nm & valid);
        }
        for (int d = 32; d > 0; d >>= 1) {
            gcnt += __shfl_down(gcnt, d);
            ncnt += __shfl_down(ncnt, d);
        }
        if (lane == 0) {
            red[2 * wave] = gcnt;
            red[2 * wave + 1] = ncnt;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long G = 0, N = 0;
            for (int w = 0; w < GC_THREADS / 64; w++) {
                G += red[2 * w];
                N += red[2 * w + 1];
            }
            out[bin] = gc_result(G, N, (unsigned long long)(hi - lo), n_min);
        }
        __syncthreads();
    }
}

// smallest n with (double)n/(double)bin_size > n_cutoff (Python: n/bin_size > n_cutoff, true division)
static unsigned long long gc_n_min(long long bin_size, double n_cutoff) {
    long long lo = 0, hi = bin_size + 1;  // answer in [0, bin_size+1]; bin_size+1 == never masked
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if ((double)mid / (double)bin_size > n_cutoff) hi = mid;
        else lo = mid + 1;
    }
    return (unsigned long long)lo;
}

extern "C" int tdt_gc_bins_device(tdt_ctx *ctx, const uint8_t *d_seq, int64_t len, int bin_size, double n_cutoff,
                                  int8_t *d_out) {
    if (!ctx || len < 0 || bin_size <= 0 || (len > 0 && (!d_seq || !d_out))) {
        tdt_set_error("tdt_gc_bins_device: bad argument");
        return TDT_E_ARG;
    }
    if (((uintptr_t)d_seq & 15) != 0) {
        tdt_set_error("tdt_gc_bins_device: d_seq must be 16-byte aligned");
        return TDT_E_ARG;
    }
    if (len == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    const long long nbins = (len + bin_size - 1) / bin_size;
    const unsigned long long n_min = gc_n_min(bin_size, n_cutoff);
    if (bin_size <= GC_SMALL_MAX) {
        // whole multiples of 256 bins per tile (every lane owns the same number of bins), ~64 KB of sequence,
        // tile bytes a multiple of 16 so every tile starts on a 16-byte boundary
        int tile_bins = 65536 / bin_size;
        tile_bins = tile_bins >= GC_THREADS ? tile_bins / GC_THREADS * GC_THREADS : (tile_bins & ~15);
        if (tile_bins < 16) tile_bins = 16;
        const long long tiles = (nbins + tile_bins - 1) / tile_bins;
        const int chunks = (tile_bins * bin_size) >> 4;
        const size_t lds = (size_t)((chunks + 1) / 2) * 2 * sizeof(unsigned);
        long long grid = tiles;
        const long long cap = (long long)ctx->num_cu * 256;   // in effect one tile per workgroup: measured 7 % faster than 16 persistent workgroups per CU
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(gc_small_bins, dim3((unsigned)grid), dim3(GC_THREADS), lds, ctx->stream, d_seq, (long long)len,
                           bin_size, tile_bins, nbins, n_min, (signed char *)d_out);
    } else {
        long long grid = nbins;
        const long long cap = (long long)ctx->num_cu * 16;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(gc_large_bins, dim3((unsigned)grid), dim3(GC_THREADS), 0, ctx->stream, d_seq, (long long)len,
                           (long long)bin_size, nbins, n_min, (signed char *)d_out);
    }
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_gc_bins(tdt_ctx *ctx, const uint8_t *seq, int64_t len, int bin_size, double n_cutoff, int8_t *out) {
    if (!ctx || len < 0 || bin_size <= 0 || (len > 0 && (!seq || !out))) {
        tdt_set_error("tdt_gc_bins: bad argument");
        return TDT_E_ARG;
    }
    if (len == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    const size_t nbins = (size_t)((len + bin_size - 1) / bin_size);
    void *d_seq = nullptr, *d_out = nullptr;
    int rc = tdt_scratch(ctx, 1, (size_t)len + 16, &d_seq);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 2, nbins, &d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_seq, seq, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    rc = tdt_gc_bins_device(ctx, (const uint8_t *)d_seq, len, bin_size, n_cutoff, (int8_t *)d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(out, d_out, nbins, hipMemcpyDeviceToHost, ctx->stream));
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    return TDT_OK;
}

// FASTA-layout variants: `raw` = the contig's bytes exactly as in the file (nbytes of them, line ends included), `len` bases,
// `linebases` bases per full line, `linewidth` bytes per full line (the .fai columns).
extern "C" int tdt_gc_bins_fasta_device(tdt_ctx *ctx, const uint8_t *d_raw, int64_t nbytes, int64_t len, int linebases, int linewidth,
                                        int bin_size, double n_cutoff, int8_t *d_out) {
    if (!ctx || len < 0 || nbytes < 0 || bin_size <= 0 || (len > 0 && (!d_raw || !d_out))) {
        tdt_set_error("tdt_gc_bins_fasta_device: bad argument");
        return TDT_E_ARG;
    }
    if (len == 0) return TDT_OK;
    if (linebases <= 0 || linewidth < linebases || linewidth > linebases + 2 || bin_size > GC_SMALL_MAX || len >= (1ll << 31) ||
        ((uintptr_t)d_raw & 15) != 0) {
        tdt_set_error("tdt_gc_bins_fasta_device: unsupported layout (linebases %d, linewidth %d, bin %d, len %lld): strip the line ends "
                      "on the host and use tdt_gc_bins", linebases, linewidth, bin_size, (long long)len);
        return TDT_E_UNSUPPORTED;
    }
    const long long nfull = len / linebases, tail = len - nfull * linebases;
    const long long need = nfull * linewidth + tail - (tail == 0 ? (linewidth - linebases) : 0);
    if (nbytes < need) {
        tdt_set_error("tdt_gc_bins_fasta_device: %lld bytes cannot hold %lld bases at %d/%d per line", (long long)nbytes, (long long)len,
                      linebases, linewidth);
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(ctx->device));
    // floor(b / linebases) = mulhi(b, magic) >> shift for 0 <= b < 2^31 (round-up method, as for the coverage bins)
    unsigned magic = 0;
    int shift = -1;
    if (linebases > 1) {
        int l = 0;
        while ((1ll << l) < linebases) l++;
        const unsigned long long m = ((1ull << (31 + l)) + (unsigned long long)linebases - 1) / (unsigned long long)linebases;
        if (m > 0xffffffffull) {
            tdt_set_error("tdt_gc_bins_fasta_device: line length %d has no 32-bit reciprocal", linebases);
            return TDT_E_UNSUPPORTED;
        }
        magic = (unsigned)m;
        shift = l - 1;   // mulhi already drops 32 bits: (b * m) >> (31 + l) == mulhi(b, m) >> (l - 1)
    }
    const long long nbins = (len + bin_size - 1) / bin_size;
    const unsigned long long n_min = gc_n_min(bin_size, n_cutoff);
    int tile_bins = 65536 / bin_size;
    tile_bins = tile_bins >= GC_THREADS ? tile_bins / GC_THREADS * GC_THREADS : (tile_bins & ~15);
    if (tile_bins < 16) tile_bins = 16;
    const long long tile_bases = (long long)tile_bins * bin_size;
    const long long tile_span = (tile_bases / linebases + 2) * linewidth + 32;        // bytes a tile can cover, generously
    const int max_chunks = (int)((tile_span + 15) >> 4);
    const size_t lds = (size_t)((max_chunks + 1) / 2) * 2 * sizeof(unsigned);
    long long grid = (nbins + tile_bins - 1) / tile_bins;
    const long long cap = (long long)ctx->num_cu * 256;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(gc_fasta_bins, dim3((unsigned)grid), dim3(GC_THREADS), lds, ctx->stream, d_raw, (long long)nbytes, (long long)len, bin_size,
                       (unsigned)linebases, (unsigned)linewidth, magic, shift, tile_bins, max_chunks, nbins,
                       n_min > 0xffffffffull ? 0xffffffffu : (unsigned)n_min, (signed char *)d_out);
    TDT_CHECK_LAUNCH();
    return TDT_OK;
}

extern "C" int tdt_gc_bins_fasta(tdt_ctx *ctx, const uint8_t *raw, int64_t nbytes, int64_t len, int linebases, int linewidth, int bin_size,
                                 double n_cutoff, int8_t *out) {
    if (!ctx || len < 0 || nbytes < 0 || bin_size <= 0 || (len > 0 && (!raw || !out))) {
        tdt_set_error("tdt_gc_bins_fasta: bad argument");
        return TDT_E_ARG;
    }
    if (len == 0) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    const size_t nbins = (size_t)((len + bin_size - 1) / bin_size);
    void *d_raw = nullptr, *d_out = nullptr;
    int rc = tdt_scratch(ctx, 1, (size_t)nbytes + 16, &d_raw);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 2, nbins, &d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_raw, raw, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = tdt_gc_bins_fasta_device(ctx, (const uint8_t *)d_raw, nbytes, len, linebases, linewidth, bin_size, n_cutoff, (int8_t *)d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(out, d_out, nbins, hipMemcpyDeviceToHost, ctx->stream));
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    return TDT_OK;
}

// Many contigs in one call: `raw` holds the FASTA bytes of n contigs, contig i at raw_off[i] (a multiple of 16: the kernel reads 16-byte
// chunks) for raw_len[i] bytes; its bins go to out + out_off[i].  One copy in, one kernel launch per contig behind it, one copy out, ONE wait
// — tiddit_gc.main's loop over the contigs of a human reference (tiddit_gc.pyx:35-42: 3 366 of them with the alt / decoy / HLA contigs) paid a
// host-device round trip per contig: 0.2 s for 25 Mb of sequence.
extern "C" int tdt_gc_bins_fasta_many(tdt_ctx *ctx, const uint8_t *raw, int64_t nbytes, int n, const int64_t *raw_off, const int64_t *raw_len,
                                      const int64_t *len, const int32_t *linebases, const int32_t *linewidth, int bin_size, double n_cutoff,
                                      int8_t *out, const int64_t *out_off, int64_t out_bytes) {
    if (!ctx || n < 0 || nbytes < 0 || out_bytes < 0 || bin_size <= 0 || (n && (!raw_off || !raw_len || !len || !linebases || !linewidth || !out_off)) ||
        (nbytes && !raw) || (out_bytes && !out)) {
        tdt_set_error("tdt_gc_bins_fasta_many: bad argument");
        return TDT_E_ARG;
    }
    for (int i = 0; i < n; i++) {
        const int64_t nb = len[i] > 0 ? (len[i] + bin_size - 1) / bin_size : 0;
        if (len[i] < 0 || raw_off[i] < 0 || raw_len[i] < 0 || (raw_off[i] & 15) || raw_off[i] + raw_len[i] > nbytes || out_off[i] < 0 || out_off[i] + nb > out_bytes) {
            tdt_set_error("tdt_gc_bins_fasta_many: contig %d lies outside the buffers (or its bytes do not start at a multiple of 16)", i);
            return TDT_E_ARG;
        }
    }
    if (!n || !out_bytes) return TDT_OK;
    TDT_HIP(hipSetDevice(ctx->device));
    void *d_raw = nullptr, *d_out = nullptr;
    int rc = tdt_scratch(ctx, 1, (size_t)nbytes + 16, &d_raw);
    if (rc) return rc;
    rc = tdt_scratch(ctx, 2, (size_t)out_bytes, &d_out);
    if (rc) return rc;
    TDT_HIP(hipMemcpyAsync(d_raw, raw, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
    for (int i = 0; i < n; i++) {
        if (len[i] == 0) continue;
        rc = tdt_gc_bins_fasta_device(ctx, (const uint8_t *)d_raw + raw_off[i], raw_len[i], len[i], linebases[i], linewidth[i], bin_size, n_cutoff,
                                      (int8_t *)d_out + out_off[i]);
        if (rc) return rc;
    }
    TDT_HIP(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    TDT_HIP(hipStreamSynchronize(ctx->stream));
    return TDT_OK;
}
