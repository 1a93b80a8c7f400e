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

enum { TDT_NSCRATCH = 24, TDT_NPINNED = 4 };

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
