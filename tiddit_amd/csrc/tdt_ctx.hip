// Context, error reporting and scratch memory for libtiddit_hip.so.
#include "tdt_common.h"

static thread_local char g_err[1024] = "";

void tdt_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *tdt_last_error(void) { return g_err; }
extern "C" int tdt_version(void) { return 100; }

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
    for (int i = 0; i < 4; i++) TDT_HIP(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
    TDT_HIP(hipMalloc((void **)&c->d_async_err, 64));
    TDT_HIP(hipMemset(c->d_async_err, 0, 64));
    c->stream = c->own_stream;
    *out = c;
    return TDT_OK;
}

extern "C" void tdt_ctx_destroy(tdt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy_stream);
    if (c->d_async_err) (void)hipFree(c->d_async_err);
    for (auto &b : c->scratch)
        if (b.p) (void)hipFree(b.p);
    for (auto &b : c->pinned)
        if (b.p) (void)hipHostFree(b.p);
    for (int i = 0; i < 4; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
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
        if (hipMalloc(&b.p, want) != hipSuccess) {
            (void)hipGetLastError();
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

