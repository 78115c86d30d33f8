// mock_rccl.cpp -- TEST INFRASTRUCTURE (tests/test_dist_gpu.py): a stand-in for librccl that lets SEVERAL ranks share ONE GPU.
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), so on a one-GPU box the path's own exchange (gf_dist_*: the per-level
// gradient-segment all-reduces inside gf_smp_backward, GraphFlow/SMP_omega.h:750-792) could only ever run with a world of one.  This
// library exports the handful of entry points gf_dist.hip binds (GF_RCCL_LIBRARY selects it) and carries the collectives through a
// shared-memory segment between the rank processes: a collective waits for its stream, stages the buffer on the host, meets the other
// ranks at a barrier, sums the ranks' slots IN RANK ORDER and copies the result back on the stream.  It exercises OUR side of the
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
constexpr size_t kSlotBytes = 16u << 20;   // per rank: 4 M floats (the SMP gradient is 0.9 MB)
constexpr int kMaxRanks = 8;
struct Header {
    std::atomic<int> joined, arrived, generation, left;
};
struct Comm {
    int rank, world;
    char name[80];
    Header *hdr;
    char *slots;
    size_t bytes;
    float *host;   // pinned staging of this rank
};
bool barrier(Comm *c, double limit_s = 120.0) {
    const int gen = c->hdr->generation.load();
    if (c->hdr->arrived.fetch_add(1) + 1 == c->world) {
        c->hdr->arrived.store(0);
        c->hdr->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->generation.load() == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return false;
    }
    return true;
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
    if (hipHostMalloc(reinterpret_cast<void **>(&c->host), kSlotBytes, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    c->hdr->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();   // ncclCommInitRank returns when every rank has called it
    while (c->hdr->joined.load() < nranks) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3600.0) return ncclSystemError;
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

static ncclResult_t release(ncclComm_t comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    const bool last = c->hdr->left.fetch_add(1) + 1 == c->world;
    munmap(c->hdr, c->bytes);
    if (last) shm_unlink(c->name);
    (void)hipHostFree(c->host);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return release(comm); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return release(comm); }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (dt != ncclFloat32 || op != ncclSum || count * sizeof(float) > kSlotBytes) return ncclInvalidArgument;
    if (hipMemcpyAsync(c->host, send, count * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;   // (everything the stream was told to wait for included)
    std::memcpy(c->slots + (size_t)c->rank * kSlotBytes, c->host, count * sizeof(float));
    if (!barrier(c)) return ncclSystemError;
    for (size_t i = 0; i < count; ++i) {   // rank order: every rank forms the same sum
        float s = 0.f;
        for (int r = 0; r < c->world; ++r) s += reinterpret_cast<const float *>(c->slots + (size_t)r * kSlotBytes)[i];
        c->host[i] = s;
    }
    if (!barrier(c)) return ncclSystemError;   // (nobody overwrites a slot before everybody has read it)
    if (hipMemcpyAsync(recv, c->host, count * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;   // (the staging buffer is reused by the next call)
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (dt != ncclFloat32 || count * sizeof(float) > kSlotBytes || root < 0 || root >= c->world) return ncclInvalidArgument;
    if (c->rank == root) {
        if (hipMemcpyAsync(c->host, send, count * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
        std::memcpy(c->slots + (size_t)root * kSlotBytes, c->host, count * sizeof(float));
    } else if (hipStreamSynchronize(stream) != hipSuccess) {
        return ncclUnhandledCudaError;
    }
    if (!barrier(c)) return ncclSystemError;
    std::memcpy(c->host, c->slots + (size_t)root * kSlotBytes, count * sizeof(float));
    if (!barrier(c)) return ncclSystemError;
    if (hipMemcpyAsync(recv, c->host, count * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

}  // extern "C"
