// TEST INFRASTRUCTURE — a stand-in for librccl.so that lets N processes sharing ONE GPU run the product's RCCL code path
// (csrc/tdt_comm.hip: tdt_comm_init / tdt_allgatherv / tdt_allreduce_sum_f64) with N > 1.  The GPU boxes of this project have one
// device, and RCCL refuses two ranks on the same device; this library implements just the entry points tdt_comm.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart/End, ncclBroadcast, ncclAllReduce, ncclGetErrorString) over a
// POSIX shared-memory segment + hipMemcpy: a collective synchronises the caller's stream, stages through host memory and meets the
// other ranks at a sense-reversing barrier.  Same call sequence, same displacements, same stream ordering contract (the result is
// complete when the call returns, which is stronger than RCCL's "complete in stream order").  Selected with TIDDIT_RCCL_LIB=<this .so>.
// Not shipped, not linked into libtiddit_hip.so; built by tests/test_gpu_comm.py with hipcc.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <ctime>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr size_t CHUNK = 4u << 20;            // bytes staged per rank per step
constexpr int MAX_RANKS = 16;

struct Control {
    std::atomic<unsigned> ready;              // 0x600d once rank 0 has initialised the block
    std::atomic<unsigned> arrived, sense;
};

struct Comm {
    char name[64];
    int rank, world;
    unsigned local_sense;
    Control *ctl;
    unsigned char *data;                      // MAX_RANKS * CHUNK bytes behind the control block
    size_t map_len;
};

void barrier(Comm *c) {
    const unsigned s = c->local_sense ^= 1u;
    if (c->ctl->arrived.fetch_add(1) + 1 == (unsigned)c->world) {
        c->ctl->arrived.store(0);
        c->ctl->sense.store(s);
    } else {
        while (c->ctl->sense.load() != s) usleep(20);
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tdt_rccl_standin_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->rank = rank;
    c->world = nranks;
    c->local_sense = 0;
    c->map_len = 4096 + (size_t)MAX_RANKS * CHUNK;
    int fd = -1;
    for (int tries = 0; tries < 20000 && fd < 0; tries++) {            // rank 0 creates, the others wait for it
        fd = rank == 0 ? shm_open(c->name, O_CREAT | O_RDWR, 0600) : shm_open(c->name, O_RDWR, 0600);
        if (fd < 0) usleep(500);
    }
    if (fd < 0) {
        delete c;
        return ncclSystemError;
    }
    if (rank == 0 && ftruncate(fd, (off_t)c->map_len) != 0) {
        close(fd);
        delete c;
        return ncclSystemError;
    }
    if (rank != 0) {                                                   // the segment has its size once rank 0 has truncated it
        for (int tries = 0; tries < 20000; tries++) {
            const off_t len = lseek(fd, 0, SEEK_END);
            if (len >= (off_t)c->map_len) break;
            usleep(500);
        }
    }
    void *p = mmap(nullptr, c->map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->ctl = (Control *)p;
    c->data = (unsigned char *)p + 4096;
    if (rank == 0) {
        c->ctl->arrived.store(0);
        c->ctl->sense.store(0);
        c->ctl->ready.store(0x600du);
    } else {
        while (c->ctl->ready.load() != 0x600du) usleep(100);
    }
    barrier(c);
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    barrier(c);
    munmap((void *)c->ctl, c->map_len);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (stand-in)" : "system error (stand-in)"; }

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || root < 0 || root >= c->world || (datatype != ncclUint8 && datatype != ncclInt8 && datatype != ncclChar)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t off = 0; off < count; off += CHUNK) {
        const size_t n = count - off < CHUNK ? count - off : CHUNK;
        if (c->rank == root && hipMemcpy(c->data, (const char *)sendbuff + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        barrier(c);
        if (c->rank != root || sendbuff != recvbuff)
            if (hipMemcpy((char *)recvbuff + off, c->data, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        barrier(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || datatype != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const size_t per = CHUNK / 8;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        double *mine = (double *)(c->data + (size_t)c->rank * CHUNK);
        if (hipMemcpy(mine, (const double *)sendbuff + off, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        barrier(c);
        double *sum = new double[n];
        for (size_t i = 0; i < n; i++) {
            double s = 0;
            for (int r = 0; r < c->world; r++) s += ((const double *)(c->data + (size_t)r * CHUNK))[i];      // rank order on every rank: identical sums
            sum[i] = s;
        }
        const hipError_t e = hipMemcpy((double *)recvbuff + off, sum, n * 8, hipMemcpyHostToDevice);
        delete[] sum;
        if (e != hipSuccess) return ncclUnhandledCudaError;
        barrier(c);
    }
    return ncclSuccess;
}
}
