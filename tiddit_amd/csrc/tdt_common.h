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
    hipStream_t back_stream = nullptr;   // device-to-host copies that must not queue behind the launch stream's next kernels (tdt_signal_scan_result)
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

// Device allocations of the library outside the ingest's own buffers: hipMalloc, and where that fails once more after the ingest's
// buffer cache (tdt_ingest.hip) has gone back to the driver — the cache must never be what makes another allocator fail.
hipError_t tdt_dev_malloc(void **p, size_t bytes);
size_t tdt_dev_cache_flush(int device);      // device < 0: every device; returns the bytes released (callers hold no cached pointer)
size_t tdt_dev_cache_held(int device);
int tdt_scratch(tdt_ctx *ctx, int slot, size_t bytes, void **out);
int tdt_pinned(tdt_ctx *ctx, int slot, size_t bytes, void **out);

static inline int tdt_ceil_log2_u64(uint64_t v) {
    int l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

#include "tdt_cov_record.h"      // the 8-byte coverage records (packed / binned): their own header, so that the profiles know when they changed

// ---- BGZF block table shared by the host scan (tdt_bgzf.hip), the device inflate (tdt_inflate.hip) and the ingest (tdt_ingest.hip)
struct BzDesc {
    unsigned long long in_off, out_off;   // payload offset in the compressed buffer, block offset in the output
    unsigned in_len, isize, crc, pad;
};
// tdt_split_fields (tdt_format.hip) / the signal tables (tdt_sigtab.hip)
struct TdtSplitOut {
    int32_t status;        // 0: SA mapQ below min_q (no row, :40-41), 1: fields valid, 2: not handled here
    int32_t read_start, read_end;      // reference_start + 1, reference_end + 1 of the read (:60-61)
    int32_t split_pos, sa_split;       // before the swap (:62-116)
    int32_t seg_start, seg_end;        // the SA segment's reference_start (the tag's POS, stored raw, :13) and reference_end
    uint32_t chr_off, chr_len;         // the SA contig name inside raw
    uint8_t is_reverse, sa_minus, pad[2];
};
void tdt_split_one(const uint8_t *meta, const uint32_t *raw_end, const uint8_t *raw, size_t raw_len, uint32_t r, int min_q, TdtSplitOut &o);

int tdt_host_thread_count();
int tdt_bz_hop(const uint8_t *p, size_t avail, size_t *bsize, size_t *pay_off, size_t *pay_len, uint32_t *isize);
int tdt_bz_block_table(const uint8_t *comp, size_t len, std::vector<BzDesc> &blocks, size_t *produced);
int tdt_bz_launch(tdt_ctx *ctx, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks, unsigned char *d_out, bool check_crc,
                  unsigned *d_status, unsigned *d_summary);
int tdt_bz_launch_on(tdt_ctx *ctx, hipStream_t st, int reserve, const unsigned char *d_comp, const BzDesc *d_blocks, size_t nblocks,
                     unsigned char *d_out, bool check_crc, unsigned *d_status, unsigned *d_summary);
const char *tdt_bz_err_name(unsigned e);
