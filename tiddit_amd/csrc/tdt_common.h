// Internal shared definitions for libtiddit_hip.so (gfx950 only; no portability layer).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/tiddit_hip.h"

#define TDT_WAVE 64

void tdt_set_error(const char *fmt, ...);

#define TDT_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            tdt_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return TDT_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define TDT_CHECK_LAUNCH() TDT_HIP(hipGetLastError())

// Grow-only device / pinned-host scratch buffers owned by a context.
struct tdt_buf {
    void *p = nullptr;
    size_t cap = 0;
};

enum { TDT_NSCRATCH = 28, TDT_NPINNED = 4 };

struct tdt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;    // stream in use (own_stream or an adopted one)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int num_cu = 256;
    tdt_buf scratch[TDT_NSCRATCH];
    tdt_buf pinned[TDT_NPINNED];
    int *d_async_err = nullptr;  // device word kernels OR into (bounded-spin timeouts); checked by tdt_ctx_sync
    unsigned tile_calls = 0;             // clustering: parity selects one of two group-sum arrays
    int tile_groups_max = 0;             // ... and how many of their entries have ever been used
    void *tile_flags_zeroed = nullptr;   // clustering: the status block that has been zeroed once (its users re-zero it themselves)
    // tdt_signal_scan -> tdt_signal_scan_result: what the last scan of THIS context selected (pointers into its scratch slots)
    size_t scan_n_sel = 0, scan_raw_bytes = 0;
    void *scan_meta = nullptr, *scan_size = nullptr, *scan_bytes = nullptr;
};

int tdt_scratch(tdt_ctx *ctx, int slot, size_t bytes, void **out);
int tdt_pinned(tdt_ctx *ctx, int slot, size_t bytes, void **out);

static inline int tdt_ceil_log2_u64(uint64_t v) {
    int l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

// ---- packed alignment record of the coverage path (8 B instead of 11), written by the ingest kernel next to the field arrays:
//   low word  = reference_start (int32)
//   high word = span:24 | min(mapq,63):6 | unmapped(0x4):1 | duplicate(0x400):1     span = reference_end - reference_start;
//               span 0xffffff = escape: the true end is read from the `end` array (reads spanning >= 16 Mb)
#define COV_PK_SPAN 0xffffffu
__host__ __device__ __forceinline__ unsigned long long cov_pack_record(int start, int end, unsigned mapq, unsigned flag) {
    const long long span = (long long)end - (long long)start;
    const unsigned sp = (span < 0 || span >= (long long)COV_PK_SPAN) ? COV_PK_SPAN : (unsigned)span;
    const unsigned info = sp | ((mapq > 63u ? 63u : mapq) << 24) | ((flag & 0x4u) ? 1u << 30 : 0u) | ((flag & 0x400u) ? 1u << 31 : 0u);
    return ((unsigned long long)info << 32) | (unsigned)start;
}

// ---- BINNED alignment record (8 B) for ONE histogram's bin size: everything about a read that does not depend on where the
// accumulating workgroup's window is, precomputed when the record is written (the ingest kernel, or tdt_cov_pack_binned_device):
//   low word  = first_bin << 2 | shape      first_bin = reference_start // bin_size (tiddit_coverage.pyx:50), clamped to the contig's bins
//       shape 0 (COV_BN_SINGLE)  the read lies in one bin (:53-57, always bases / bin_size — also in the contig's last bin)
//       shape 1 (COV_BN_MULTI)   first and last bin differ and the last bin is NOT the contig's last one (:59-66); histograms with
//                                bins >= 129 bp (cov_accumulate MODE 0) take only reads of exactly two bins here, the small-bin
//                                flavour (MODE 1) up to 255 bins after the first
//       shape 2 (COV_BN_SLOW)    any other valid read (more bins, or ending in the contig's last bin: other denominator, :67-69):
//                                replayed literally from the start / end arrays
//       shape 3 (COV_BN_INVALID) start < 0, end <= start or a bin beyond the contig: the reference raises IndexError when such a
//                                read passes the filter
//   high word = duplicate:1 | unmapped:1 | min(mapq,63):6 (top byte, as in cov_pack_record: the read filter is one range test)
//               | table indices: MODE 0  bases_last_bin:10 | bases_first_bin:10 | 0000   (the first-bin index already scaled to the
//                                        16-byte table entries: the accumulation launch masks it out with one AND)
//                                MODE 1  bins_after_first:8 | bases_last_bin:8 | bases_first_bin:8
//   bases_first_bin = end - start (single) or (first_bin+1)*bin_size - start; bases_last_bin = (end-1) - end_bin*bin_size, the
//   reference's one-short count (:63).
#define COV_BN_SINGLE 0u
#define COV_BN_MULTI 1u
#define COV_BN_SLOW 2u
#define COV_BN_INVALID 3u
struct CovBinSpec {
    unsigned z = 0;            // bin size; 0 = no binned records for this histogram (bin_size 1 or >= 1024)
    unsigned magic = 0;        // floor(x / z) = mulhi(x, magic) >> shift for 0 <= x < 2^31
    int shift = 0;
    int mode1 = 0;             // the small-bin kernel flavour's field layout
    const int *d_nbins = nullptr;   // bins of every contig (device array, by contig id)
    int n_contigs = 0;
};
__host__ __device__ __forceinline__ unsigned long long cov_bin_record(int start, int end, unsigned mapq, unsigned flag, int nbins, unsigned z,
                                                                      unsigned magic, int shift, bool mode1) {
    const unsigned top = ((mapq > 63u ? 63u : mapq) << 24) | ((flag & 0x4u) ? 1u << 30 : 0u) | ((flag & 0x400u) ? 1u << 31 : 0u);
    const int last_bin = nbins - 1;
    auto div = [&](int x) -> int { return (int)((unsigned)(((unsigned long long)(unsigned)x * magic) >> 32) >> shift); };
    unsigned shape = COV_BN_INVALID, fields = 0;
    int fb = 0;
    if (start >= 0 && end > start && nbins > 0) {
        fb = div(start);
        const int eb = div(end - 1);
        if (eb > last_bin) {
            fb = fb > last_bin ? last_bin : fb;
        } else if (eb == fb) {
            shape = COV_BN_SINGLE;
            fields = mode1 ? (unsigned)(end - start) : (unsigned)(end - start) << 4;
        } else if (eb >= last_bin || eb - fb > (mode1 ? 255 : 1)) {
            shape = COV_BN_SLOW;
        } else {
            shape = COV_BN_MULTI;
            const unsigned bf = (unsigned)(fb + 1) * z - (unsigned)start, bl = (unsigned)(end - 1) - (unsigned)eb * z;
            fields = mode1 ? (bf | (bl << 8) | ((unsigned)(eb - fb) << 16)) : ((bf << 4) | (bl << 14));
        }
    } else if (start >= 0 && nbins > 0) {
        fb = div(start);
        fb = fb > last_bin ? last_bin : fb;
    }
    return ((unsigned long long)(top | fields) << 32) | ((unsigned)fb << 2) | shape;
}

// ---- BGZF block table shared by the host scan (tdt_bgzf.hip), the device inflate (tdt_inflate.hip) and the ingest (tdt_ingest.hip)
struct BzDesc {
    unsigned long long in_off, out_off;   // payload offset in the compressed buffer, block offset in the output
    unsigned in_len, isize, crc, pad;
};
int tdt_host_thread_count();
int tdt_bz_hop(const uint8_t *p, size_t avail, size_t *bsize, size_t *pay_off, size_t *pay_len, uint32_t *isize);
int tdt_bz_block_table(const uint8_t *comp, size_t len, std::vector<BzDesc> &blocks, size_t *produced);
int tdt_bz_launch(tdt_ctx *ctx, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks, unsigned char *d_out, bool check_crc,
                  unsigned *d_status, unsigned *d_summary);
const char *tdt_bz_err_name(unsigned e);
