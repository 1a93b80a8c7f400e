// Multi-GPU exchange behind the C ABI: one process per GPU, RCCL over xGMI.
//   * tdt_allgatherv          the ONE exchange step of the clustering path — every rank's label array to every rank
//                             (RCCL has no allgatherv: one ncclBroadcast per rank inside a group call);
//   * tdt_allreduce_sum_f64   the exact reduction of the coverage bins when one BAM is read as byte-range shards.
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch-ROCm ships its own librccl.so) keeps
// using that one — two RCCL instances in one process do not share their state — and a plain C/C++ caller gets ROCm's.
#include "tdt_common.h"
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.h) return TDT_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // TIDDIT_RCCL_LIB=<path>: bind exactly this library (a site's own RCCL build; the tests' shared-memory stand-in that lets N ranks
    // share one GPU, tests/rccl_standin/) — RTLD_LOCAL, so that its nccl* symbols do not shadow an RCCL the process already carries
    if (const char *forced = getenv("TIDDIT_RCCL_LIB")) {
        if (!(h = dlopen(forced, RTLD_NOW | RTLD_LOCAL))) {
            tdt_set_error("TIDDIT_RCCL_LIB=%s: %s", forced, dlerror());
            return TDT_E_UNSUPPORTED;
        }
    }
    if (!h)
        for (const char *n : names)                  // an instance the process already has (e.g. PyTorch's) first
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        tdt_set_error("RCCL not found (librccl.so): %s", dlerror());
        return TDT_E_UNSUPPORTED;
    }
    Rccl r;
    r.h = h;
#define TDT_SYM(field, name)                                                      \
    r.field = (decltype(r.field))dlsym(h, name);                                  \
    if (!r.field) {                                                               \
        tdt_set_error("RCCL symbol %s missing", name);                            \
        return TDT_E_UNSUPPORTED;                                                 \
    }
    TDT_SYM(GetUniqueId, "ncclGetUniqueId")
    TDT_SYM(CommInitRank, "ncclCommInitRank")
    TDT_SYM(CommDestroy, "ncclCommDestroy")
    TDT_SYM(Broadcast, "ncclBroadcast")
    TDT_SYM(AllReduce, "ncclAllReduce")
    TDT_SYM(GroupStart, "ncclGroupStart")
    TDT_SYM(GroupEnd, "ncclGroupEnd")
    TDT_SYM(GetErrorString, "ncclGetErrorString")
#undef TDT_SYM
    g_rccl = r;
    return TDT_OK;
}
}  // namespace

#define TDT_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            tdt_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return TDT_E_HIP;                                                                       \
        }                                                                                           \
    } while (0)

struct tdt_comm {
    tdt_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int tdt_comm_unique_id(uint8_t *id128) {
    if (!id128) {
        tdt_set_error("tdt_comm_unique_id: null buffer");
        return TDT_E_ARG;
    }
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    TDT_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return TDT_OK;
}

extern "C" int tdt_comm_init(tdt_ctx *ctx, const uint8_t *id128, int rank, int world, tdt_comm **out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) {
        tdt_set_error("tdt_comm_init: bad argument");
        return TDT_E_ARG;
    }
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    TDT_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    tdt_comm *c = new tdt_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        tdt_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return TDT_E_HIP;
    }
    *out = c;
    return TDT_OK;
}

extern "C" int tdt_comm_destroy(tdt_comm *c) {
    if (!c) return TDT_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return TDT_OK;
}

// The layout of one variable-count all-gather, as a pure function of (rank, world, counts, displs): the list of broadcasts the group
// call issues — root, destination byte range in d_recv, and whether this rank's source is its send buffer (it is the root) or the
// destination itself.  Separate from the RCCL calls so that the count / displacement arithmetic is testable on the host for any
// world size (tests/test_abi.py); tdt_allgatherv executes exactly this list.
extern "C" int tdt_allgatherv_plan(int rank, int world, const size_t *counts, const size_t *displs, int elem_bytes, size_t recv_capacity_bytes,
                                   tdt_gather_op *ops, int *n_ops) {
    if (world < 1 || rank < 0 || rank >= world || !counts || !displs || elem_bytes <= 0 || !ops || !n_ops) {
        tdt_set_error("tdt_allgatherv_plan: bad argument");
        return TDT_E_ARG;
    }
    int k = 0;
    for (int r = 0; r < world; r++) {
        if (!counts[r]) continue;                                   // an empty contribution is no broadcast at all (on every rank alike)
        if (counts[r] > SIZE_MAX / (size_t)elem_bytes || displs[r] > SIZE_MAX / (size_t)elem_bytes) {
            tdt_set_error("tdt_allgatherv_plan: rank %d's range overflows size_t", r);
            return TDT_E_ARG;
        }
        const size_t off = displs[r] * (size_t)elem_bytes, nb = counts[r] * (size_t)elem_bytes;
        if (recv_capacity_bytes && (off > recv_capacity_bytes || nb > recv_capacity_bytes - off)) {
            tdt_set_error("tdt_allgatherv_plan: rank %d's range [%zu, %zu) leaves the receive buffer of %zu bytes", r, off, off + nb, recv_capacity_bytes);
            return TDT_E_ARG;
        }
        for (int q = 0; q < k; q++)                                 // two ranks' ranges must not overlap
            if (off < ops[q].offset_bytes + ops[q].nbytes && ops[q].offset_bytes < off + nb) {
                tdt_set_error("tdt_allgatherv_plan: the ranges of ranks %d and %d overlap", ops[q].root, r);
                return TDT_E_ARG;
            }
        ops[k].root = r;
        ops[k].offset_bytes = off;
        ops[k].nbytes = nb;
        ops[k].from_send = r == rank;
        k++;
    }
    *n_ops = k;
    return TDT_OK;
}

extern "C" int tdt_allgatherv(tdt_comm *c, const void *d_send, size_t send_count, void *d_recv, const size_t *counts, const size_t *displs,
                              int elem_bytes) {
    if (!c || !d_recv || !counts || !displs || elem_bytes <= 0 || (send_count && !d_send) || counts[c->rank] != send_count) {
        tdt_set_error("tdt_allgatherv: bad argument (counts[rank] must equal send_count)");
        return TDT_E_ARG;
    }
    std::vector<tdt_gather_op> ops((size_t)c->world);
    int n_ops = 0;
    int rc = tdt_allgatherv_plan(c->rank, c->world, counts, displs, elem_bytes, 0, ops.data(), &n_ops);
    if (rc) return rc;
    TDT_HIP(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    TDT_NCCL(g_rccl.GroupStart());
    for (int k = 0; k < n_ops; k++) {
        char *dst = (char *)d_recv + ops[k].offset_bytes;
        const void *src = ops[k].from_send ? d_send : (const void *)dst;
        ncclResult_t e = g_rccl.Broadcast(src, dst, ops[k].nbytes, ncclUint8, ops[k].root, c->comm, st);
        if (e != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            tdt_set_error("ncclBroadcast (root %d) failed: %s", ops[k].root, g_rccl.GetErrorString(e));
            return TDT_E_HIP;
        }
    }
    TDT_NCCL(g_rccl.GroupEnd());
    return TDT_OK;
}

extern "C" int tdt_allreduce_sum_f64(tdt_comm *c, double *d_buf, size_t n) {
    if (!c || (n && !d_buf)) {
        tdt_set_error("tdt_allreduce_sum_f64: bad argument");
        return TDT_E_ARG;
    }
    if (!n) return TDT_OK;
    TDT_HIP(hipSetDevice(c->ctx->device));
    TDT_NCCL(g_rccl.AllReduce(d_buf, d_buf, n, ncclFloat64, ncclSum, c->comm, c->ctx->stream));
    return TDT_OK;
}
