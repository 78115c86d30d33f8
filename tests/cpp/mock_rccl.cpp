// mock_rccl.cpp -- TEST INFRASTRUCTURE (tests/test_dist_gpu.py): a stand-in for librccl that lets SEVERAL ranks share ONE GPU.
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), so on a one-GPU box the path's own exchange (gf_dist_*: the per-level
// gradient-segment all-reduces inside gf_smp_backward, GraphFlow/SMP_omega.h:750-792) could only ever run with a world of one.  This
// library exports the handful of entry points gf_dist.hip binds (GF_RCCL_LIBRARY selects it) and carries the collectives through a
// shared-memory segment between the rank processes, ASYNCHRONOUSLY like the real thing: a collective enqueues, on its stream, a copy
// to pinned memory, a host function that meets the other ranks at a barrier and sums the ranks' slots IN RANK ORDER, and the copy back.  It exercises OUR side of the
// exchange -- which segments, offsets and counts, the stream choreography, the join, teardown -- not RCCL's transport.
// Build: tests/cpp/Makefile (hipcc -shared).  Never linked or loaded by the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr size_t kSlotBytes = 4u << 20;   // per rank: 1 M floats (the SMP gradient is 0.9 MB)
constexpr int kMaxRanks = 8, kRing = 8;   // staging buffers in rotation: collectives in flight per communicator
struct Header {
    std::atomic<int> joined, arrived, generation, left;
};
struct Comm {
    int rank, world;
    char name[80];
    Header *hdr;
    char *slots;
    size_t bytes;
    float *ring[kRing];   // pinned staging
    unsigned next = 0;
    std::atomic<int> aborted{0};
};
// every rank of the world, or false: this communicator was aborted (ncclCommAbort) while it waited
bool barrier(Comm *c) {
    const int gen = c->hdr->generation.load();
    if (c->hdr->arrived.fetch_add(1) + 1 == c->world) {
        c->hdr->arrived.store(0);
        c->hdr->generation.fetch_add(1);
        return true;
    }
    while (c->hdr->generation.load() == gen) {
        if (c->aborted.load()) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return true;
}
struct Op {
    Comm *c;
    bool reduce;
    int root;
    size_t count;
    float *stage;
};
// Runs ON THE STREAM (hipLaunchHostFunc), between the copy out of the send buffer and the copy into the receive buffer: like RCCL's
// kernel it holds the stream until every rank has joined the collective -- a peer that never comes leaves the stream stuck, which is what
// the library's watchdog and its abortable teardown are for.
void exchange(void *arg) {
    Op *op = static_cast<Op *>(arg);
    Comm *c = op->c;
    const size_t bytes = op->count * sizeof(float);
    if (!c->aborted.load()) {
        if (op->reduce || c->rank == op->root) std::memcpy(c->slots + (size_t)c->rank * kSlotBytes, op->stage, bytes);
        if (barrier(c)) {
            if (op->reduce) {
                for (size_t i = 0; i < op->count; ++i) {   // rank order: every rank forms the same sum
                    float s = 0.f;
                    for (int r = 0; r < c->world; ++r) s += reinterpret_cast<const float *>(c->slots + (size_t)r * kSlotBytes)[i];
                    op->stage[i] = s;
                }
            } else {
                std::memcpy(op->stage, c->slots + (size_t)op->root * kSlotBytes, bytes);
            }
            (void)barrier(c);   // (nobody overwrites a slot before everybody has read it)
        }
    }
    delete op;
}
ncclResult_t collective(Comm *c, bool reduce, int root, const void *send, void *recv, size_t count, hipStream_t stream) {
    if (count * sizeof(float) > kSlotBytes) return ncclInvalidArgument;
    if (c->aborted.load()) return ncclInvalidUsage;
    float *stage = c->ring[c->next++ % kRing];
    if (hipMemcpyAsync(stage, send, count * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, exchange, new Op{c, reduce, root, count, stage}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, stage, count * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;   // (asynchronous, as the real thing: the caller orders itself against `stream`)
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::memset(id->internal, 0, sizeof id->internal);
    std::snprintf(id->internal, sizeof id->internal, "/gf_mock_rccl_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank, c->world = nranks;
    std::snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->bytes = sizeof(Header) + 64 + (size_t)nranks * kSlotBytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;   // (every rank: same size; a fresh segment reads as zeros)
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->hdr = static_cast<Header *>(p);
    c->slots = static_cast<char *>(p) + sizeof(Header) + 64;
    for (int i = 0; i < kRing; ++i)
        if (hipHostMalloc(reinterpret_cast<void **>(&c->ring[i]), kSlotBytes, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    c->hdr->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();   // ncclCommInitRank returns when every rank has called it
    while (c->hdr->joined.load() < nranks) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3600.0) return ncclSystemError;
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// (the communicator object itself is leaked on purpose: a host function still queued on a stream may hold it)
static ncclResult_t release(Comm *c) {
    const bool last = c->hdr->left.fetch_add(1) + 1 == c->world;
    if (last) shm_unlink(c->name);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return comm ? release(reinterpret_cast<Comm *>(comm)) : ncclSuccess; }
// Abort: whatever of this communicator waits for a peer stops waiting (its streams drain), later collectives are refused
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    c->aborted.store(1);
    return release(c);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (dt != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;
    return collective(reinterpret_cast<Comm *>(comm), true, 0, send, recv, count, stream);
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (dt != ncclFloat32 || root < 0 || root >= c->world) return ncclInvalidArgument;
    return collective(c, false, root, send, recv, count, stream);
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

}  // extern "C"
