// Context, error reporting and scratch memory for libtiddit_hip.so.
#include "tdt_common.h"

#include <atomic>
#include <string>

static thread_local char g_err[1024] = "";
static std::atomic<int> live_contexts[64];

void tdt_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *tdt_last_error(void) { return g_err; }
extern "C" int tdt_version(void) { return 100; }

// The measurement macros (ablations: COV_M1X_*, B2_EXP_*, DT_PERSIST ...; tunables given on the command line: COV_WIN, B2_OCC, RS_ROUNDS ...)
// the library's translation units were compiled with — "" for the product build.  Two of the ablations produce wrong results by design;
// the Python binding refuses a library that reports any unless TIDDIT_ALLOW_VARIANT=1 (tools/ab_*.sh set it).
extern const char *const tdt_variant_coverage, *const tdt_variant_dbscan, *const tdt_variant_inflate, *const tdt_variant_inflate2,
    *const tdt_variant_ingest, *const tdt_variant_sort;
extern "C" const char *tdt_build_flags(void) {
    static const std::string all = [] {
        std::string s;
        for (const char *p : {tdt_variant_coverage, tdt_variant_dbscan, tdt_variant_inflate, tdt_variant_inflate2, tdt_variant_ingest, tdt_variant_sort}) s += p;
        return s.empty() ? s : s.substr(1);
    }();
    return all.c_str();
}

extern "C" int tdt_device_count(int *count) {
    if (!count) return TDT_E_ARG;
    *count = 0;
    TDT_HIP(hipGetDeviceCount(count));
    return TDT_OK;
}

extern "C" int tdt_ctx_create(int device, tdt_ctx **out) {
    if (!out) {
        tdt_set_error("tdt_ctx_create: out is null");
        return TDT_E_ARG;
    }
    *out = nullptr;
    int n = 0;
    TDT_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        tdt_set_error("tdt_ctx_create: device %d not present (%d visible); there is no CPU fallback", device, n);
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(device));
    tdt_ctx *c = new tdt_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    TDT_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    TDT_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    TDT_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    TDT_HIP(hipStreamCreateWithFlags(&c->back_stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) TDT_HIP(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
    TDT_HIP(tdt_dev_malloc((void **)&c->d_async_err, 64));
    TDT_HIP(hipMemset(c->d_async_err, 0, 64));
    c->stream = c->own_stream;
    if (device < 64) live_contexts[device]++;
    *out = c;
    return TDT_OK;
}

extern "C" void tdt_ctx_destroy(tdt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_stream);
    if (c->back_stream) (void)hipStreamSynchronize(c->back_stream);
    if (c->d_async_err) (void)hipFree(c->d_async_err);
    for (auto &b : c->scratch)
        if (b.p) (void)hipFree(b.p);
    for (auto &b : c->pinned)
        if (b.p) (void)hipHostFree(b.p);
    for (int i = 0; i < 4; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->back_stream) (void)hipStreamDestroy(c->back_stream);
    // the device's last context takes the ingest's cached buffers with it
    if (c->device < 64 && --live_contexts[c->device] == 0) (void)tdt_dev_cache_flush(c->device);
    delete c;
}

// test hook: the next `n` device allocations of the library behave as if the driver had refused them once (the retry behind the
// cache flush is what a test then observes); 0 switches it off
static std::atomic<int> fail_next_malloc{0};
extern "C" void tdt_debug_fail_next_malloc(int n) { fail_next_malloc = n > 0 ? n : 0; }

hipError_t tdt_dev_malloc(void **p, size_t bytes) {
    hipError_t e = hipErrorOutOfMemory;
    int pending = fail_next_malloc.load();
    while (pending > 0 && !fail_next_malloc.compare_exchange_weak(pending, pending - 1)) {
    }
    const bool injected = pending > 0;
    if (!injected) e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (tdt_dev_cache_flush(dev) == 0 && !injected) return e;        // nothing of ours was in the way (an injected refusal is always tried again)
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

extern "C" int tdt_ctx_sync(tdt_ctx *c) {
    if (!c) return TDT_E_ARG;
    TDT_HIP(hipSetDevice(c->device));
    TDT_HIP(hipStreamSynchronize(c->copy_stream));
    TDT_HIP(hipStreamSynchronize(c->stream));
    int err = 0;
    TDT_HIP(hipMemcpy(&err, c->d_async_err, 4, hipMemcpyDeviceToHost));
    if (err) {
        TDT_HIP(hipMemset(c->d_async_err, 0, 4));
        tdt_set_error("a device scan timed out waiting for a predecessor tile (internal error %d)", err);
        return TDT_E_HIP;
    }
    return TDT_OK;
}

// A helper thread of the caller that pins host memory (tdt_host_alloc) or only ever hands pointers to the library makes the
// context's device its current one first (pinned allocations belong to the calling thread's current device).
extern "C" int tdt_ctx_bind_thread(tdt_ctx *c) {
    if (!c) return TDT_E_ARG;
    TDT_HIP(hipSetDevice(c->device));
    return TDT_OK;
}

extern "C" void *tdt_ctx_stream(tdt_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int tdt_ctx_set_stream(tdt_ctx *c, void *s) {
    if (!c) return TDT_E_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return TDT_OK;
}

int tdt_scratch(tdt_ctx *c, int slot, size_t bytes, void **out) {
    tdt_buf &b = c->scratch[slot];
    if (bytes > b.cap) {
        // the old block may still be in use by enqueued kernels
        TDT_HIP(hipStreamSynchronize(c->stream));
        if (b.p) TDT_HIP(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        if (tdt_dev_malloc(&b.p, want) != hipSuccess) {
            tdt_set_error("device allocation of %zu bytes failed", want);
            return TDT_E_NOMEM;
        }
        b.cap = want;
    }
    *out = b.p;
    return TDT_OK;
}

int tdt_pinned(tdt_ctx *c, int slot, size_t bytes, void **out) {
    tdt_buf &b = c->pinned[slot];
    if (bytes > b.cap) {
        TDT_HIP(hipStreamSynchronize(c->stream));
        TDT_HIP(hipStreamSynchronize(c->copy_stream));
        if (b.p) TDT_HIP(hipHostFree(b.p));
        b.p = nullptr;
        b.cap = 0;
        if (hipHostMalloc(&b.p, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            tdt_set_error("pinned host allocation of %zu bytes failed", bytes);
            return TDT_E_NOMEM;
        }
        b.cap = bytes;
    }
    *out = b.p;
    return TDT_OK;
}

// ---- what a plain streaming read reaches on this device: the yardstick beside the 8 TB/s of the data sheet (bench.py's roofline object
// quotes both).  Every lane keeps four 16-byte loads in flight over a grid-stride walk; the XOR of what it read decides a store that
// never happens for real data, which is all that keeps the loads alive.
// blocked = 1: workgroup b streams its own contiguous 1/gridDim.x of the buffer (4 KB per step); 0: grid-stride
__global__ __launch_bounds__(256) void calib_stream_read(const uint4 *__restrict__ p, size_t n16, unsigned *__restrict__ sink, int blocked) {
    size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (blocked) {
        const size_t chunk = (n16 / gridDim.x) & ~(size_t)1023;       // whole 16-KB steps; the remainder is left unread (a measurement aid)
        p += (size_t)blockIdx.x * chunk;
        n16 = chunk;
        stride = 256;
        i = threadIdx.x;
    }
    uint4 a = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        a.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x;
        a.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
        a.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z;
        a.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
    }
    for (; i < n16; i += stride) {
        const uint4 v = p[i];
        a.x ^= v.x;
        a.y ^= v.y;
        a.z ^= v.z;
        a.w ^= v.w;
    }
    const unsigned x = a.x ^ a.y ^ a.z ^ a.w;
    if (x == 0x9e3779b9u && a.x == 0x7f4a7c15u && a.y == 0xf39cc060u) sink[0] = x;
}

extern "C" int tdt_calib_stream_read(tdt_ctx *c, const void *d_buf, size_t bytes, int reps, int workgroups_per_cu, int blocked, double *best_ms,
                                     double *mean_ms) {
    if (!c || !d_buf || bytes < (1u << 20) || reps < 1 || workgroups_per_cu < 1 || workgroups_per_cu > 256 || !best_ms || !mean_ms ||
        ((uintptr_t)d_buf & 15)) {
        tdt_set_error("tdt_calib_stream_read: bad argument (a 16-byte aligned device buffer of at least 1 MiB, reps >= 1, 1..256 workgroups per CU)");
        return TDT_E_ARG;
    }
    TDT_HIP(hipSetDevice(c->device));
    void *sink = nullptr;
    int rc = tdt_scratch(c, 27, 256, &sink);
    if (rc) return rc;
    hipEvent_t e0, e1;
    TDT_HIP(hipEventCreate(&e0));
    TDT_HIP(hipEventCreate(&e1));
    const unsigned grid = (unsigned)c->num_cu * (unsigned)workgroups_per_cu;      // (8 workgroups of four waves = eight waves per SIMD)
    double best = 1e30, sum = 0;
    for (int r = -1; r < reps; r++) {                              // (one untimed pass first)
        TDT_HIP(hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(calib_stream_read, dim3(grid), dim3(256), 0, c->stream, (const uint4 *)d_buf, bytes / 16, (unsigned *)sink, blocked ? 1 : 0);
        TDT_HIP(hipEventRecord(e1, c->stream));
        TDT_HIP(hipEventSynchronize(e1));
        float ms = 0;
        TDT_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) {
            best = ms < best ? ms : best;
            sum += ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *best_ms = best;
    *mean_ms = sum / reps;
    return TDT_OK;
}
